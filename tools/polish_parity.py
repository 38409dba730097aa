#!/usr/bin/env python3
"""Consensus parity of a whole polishing round — the GUIDED windows, as they are actually run: band guide from the NW path,
racon's quality filter and quality-weighted edges, layers as the round cuts them — against the oracle's round
(oracle.polish_round = racon::Polisher::Polish restated) on the same contigs and reads.

The draft is n_contigs contigs of 50 kb (100 windows each; racon's coverage trim shortens the two end windows of a contig,
in both implementations alike), every contig with its own 30x (hifi: 40x) reads; the device
polishes all contigs in ONE round, the oracle one contig per thread.  The layer tables of the two are bit-identical
(tests/test_gpu_polish.py), so what can differ is the window consensus: every window that differs costs at least one edit,
hence  identical windows >= windows - sum of edit distances  — the bound this tool reports — and every differing contig
is listed with its distances to the truth.
    python tools/polish_parity.py [n_contigs] [threads] [shape]     shape: q10 (the metric's config) | qual | none | hifi"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402  (the checker)
from raven_amd import hip, seqio  # noqa: E402

CONTIG = 50_000


def mutate(rng, truth, sub, ins, dele):
    L = truth.shape[0]
    u = rng.random(L)
    keep = u >= dele
    base = truth.copy()
    s = (u >= dele) & (u < dele + sub)
    base[s] = (base[s] + rng.integers(1, 4, size=int(s.sum()))) & 3
    insm = rng.random(L) < ins
    emit = keep.astype(np.int64) + insm
    seq = np.repeat(base, emit)
    off = np.cumsum(emit)
    slots = off[insm] - 1
    seq[slots] = rng.integers(0, 4, size=slots.shape[0])
    return seq.astype(np.uint8)


def revcomp(c):
    return (3 - c[::-1]).astype(np.uint8)


def make_case(rng, shape):
    hifi = shape == "hifi"
    err = (0.002, 0.0015, 0.0015) if hifi else (0.04, 0.03, 0.03)
    cov, mean_len = (40, 4000) if hifi else (30, 3000)
    truth = rng.integers(0, 4, size=CONTIG, dtype=np.uint8)
    draft = mutate(rng, truth, 0.01 * (0.2 if hifi else 1.0), 0.008 * (0.2 if hifi else 1.0), 0.008 * (0.2 if hifi else 1.0))
    reads, quals = [], []
    n_reads = cov * CONTIG // mean_len
    for _ in range(n_reads):
        ln = int(np.clip(rng.normal(mean_len, mean_len / 5), 800, CONTIG))
        b = int(rng.integers(0, CONTIG - ln + 1))
        piece = mutate(rng, truth[b:b + ln], *err)
        if rng.random() < 0.5:
            piece = revcomp(piece)
        reads.append(piece)
        if shape == "q10":
            quals.append(np.full(len(piece), 33 + 10, dtype=np.uint8))
        elif shape == "qual":
            quals.append((33 + rng.integers(5, 41, size=len(piece))).astype(np.uint8))
    return truth, draft, reads, (quals if shape in ("q10", "qual") else None)


def run(n_contigs=20, threads=None, shape="q10", seed=20261001):
    rng = np.random.default_rng(seed)
    cases = [make_case(rng, shape) for _ in range(n_contigs)]
    eng = hip.Engine(15, 5)
    eng.set_option("poa_rows_min_windows", 0)  # the chain of a full-size round (rows-on-lanes kernel first) whatever the batch size
    targets = eng.upload(seqio.pack_reads([c[1] for c in cases]))
    all_reads = [r for c in cases for r in c[2]]
    all_quals = [q for c in cases for q in c[3]] if cases[0][3] is not None else None
    reads = eng.upload(seqio.pack_reads(all_reads))
    t0 = time.time()
    cons, ratio, st = eng.polish_round(targets, reads, quals=all_quals, q=10.0 if shape in ("q10", "qual") else 0.0)
    t_dev = time.time() - t0

    def one(c):
        tr, dr, rd, ql = c
        return oracle.polish_round(seqio.pack_reads([dr]), seqio.pack_reads(rd), quals=ql, q=10.0 if shape in ("q10", "qual") else 0.0)[0][0]

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=threads or os.cpu_count() or 1) as ex:  # the oracle releases the GIL inside its C++ call
        refs = list(ex.map(one, cases))
    t_cpu = time.time() - t0
    same, diffs = 0, []
    for i, (c, r) in enumerate(zip(cons, refs)):
        if np.array_equal(c, r):
            same += 1
            continue
        tr = cases[i][0]
        diffs.append({"contig": i, "ed_device_vs_oracle": int(oracle.edit_distance(bytes(c + 65), bytes(r + 65))),
                      "ed_device_vs_truth": int(oracle.edit_distance(bytes(c + 65), bytes(tr + 65))),
                      "ed_oracle_vs_truth": int(oracle.edit_distance(bytes(r + 65), bytes(tr + 65)))})
    windows = int(st["n_windows"])
    ed_sum = int(sum(d["ed_device_vs_oracle"] for d in diffs))
    return {"shape": shape, "contigs": n_contigs, "windows": windows, "layers": int(st["n_layers"]),
            "dropped_layers_by_quality": int(st["n_dropped_layers"]), "polished_windows": int(st["n_polished_windows"]),
            "failed_windows": int(st["n_failed_windows"]), "contigs_identical": same, "contigs_different": len(diffs),
            "sum_ed_device_vs_oracle": ed_sum, "max_ed_in_a_contig": int(max([d["ed_device_vs_oracle"] for d in diffs] or [0])),
            "identical_windows_at_least": windows - ed_sum, "identical_fraction_at_least": round(1.0 - ed_sum / max(windows, 1), 6),
            "sum_ed_device_vs_truth": int(sum(d["ed_device_vs_truth"] for d in diffs)),
            "sum_ed_oracle_vs_truth": int(sum(d["ed_oracle_vs_truth"] for d in diffs)),
            "device_s": round(t_dev, 2), "oracle_s": round(t_cpu, 1), "seed": seed, "different": diffs[:40]}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    th = int(sys.argv[2]) if len(sys.argv) > 2 else None
    shape = sys.argv[3] if len(sys.argv) > 3 else "q10"
    print(json.dumps(run(n, th, shape)))
