#!/usr/bin/env python3
"""Consensus parity of the window-consensus stage against the repo's own POA oracle (racon Window::GenerateConsensus over
spoa, oracle/poa_oracle.cpp) on C4-like windows: 500-base backbone with ~2.6 % errors, ~30 ONT-like layers (4 % sub, 3 % ins,
3 % del), a share of them partial.  Reports the fraction of windows whose consensus is byte-identical and, for the rest, the
edit distance between the two and of each to the truth, and whether the oracle's STATEMENT OF THE DEVICE'S RULES gives the
device's consensus: oracle.poa_window(device_order=True, end_tie=1) = same graph rules and scores, the graph's rows in the
device's incremental order (another valid topological order) and the end node of an alignment taken by smallest node id
among equal scores (round 5; spoa: first in its DFS rank) — i.e. whether the difference is such a tie and nothing else.
    python tools/poa_parity.py [n_windows] [threads] [mode] [shape]
shape (round 6: the shapes that are actually run, not only the one of rounds 4-5):
    ont    unit weights, ~10 % errors, Poisson(31) layers                                  (rounds 4-5)
    qual   the same reads with per-base qualities, Phred 5..40 at random (quality-weighted edges)
    q10    the same reads with Phred 10 throughout = the block qualities of the bench's FASTQ variant (the metric's config)
    hifi   0.5 % errors, Poisson(41) layers, unit weights                                   (configs[4]'s reads)
The band guide of rvn_poa_consensus_batch is the straight line (no alignment path is handed in): these are the UNGUIDED
windows of DESIGN.md 3.6; the guided form — the windows of a real polishing round, guide from the NW path, with the
round's quality filter — is measured by tools/polish_parity.py."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402  (the checker)
from raven_amd import hip  # noqa: E402


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


SHAPES = {  # (sub, ins, del) of a layer, mean layer count, backbone error scale, qualities
    "ont": ((0.04, 0.03, 0.03), 31, 1.0, None),
    "qual": ((0.04, 0.03, 0.03), 31, 1.0, "random"),
    "q10": ((0.04, 0.03, 0.03), 31, 1.0, "ten"),
    "hifi": ((0.002, 0.0015, 0.0015), 41, 0.2, None),
}


def make_window(rng, partial_share=0.2, shape="ont"):
    err, mean_layers, bb_scale, qual = SHAPES[shape]
    truth = rng.integers(0, 4, size=500, dtype=np.uint8)
    bb = mutate(rng, truth, 0.01 * bb_scale, 0.008 * bb_scale, 0.008 * bb_scale)
    layers, begins, ends = [bb], [0], [len(bb) - 1]
    quals = [np.full(len(bb), 33, dtype=np.uint8)]  # (the backbone is a draft contig: racon's dummy quality '!', weight 0)
    n_layers = int(np.clip(rng.poisson(mean_layers), 3, 90))
    for _ in range(n_layers):
        if rng.random() < partial_share:
            b = int(rng.integers(0, 250))
            e = int(rng.integers(b + 125, 500))
        else:
            b, e = 0, 500
        piece = mutate(rng, truth[b:e], *err)
        if len(piece) < 10:
            continue
        layers.append(piece)
        if qual == "random":
            quals.append((33 + rng.integers(5, 41, size=len(piece))).astype(np.uint8))
        elif qual == "ten":
            quals.append(np.full(len(piece), 33 + 10, dtype=np.uint8))
        bb_b = min(len(bb) - 2, int(b * len(bb) / 500))
        bb_e = min(len(bb) - 1, max(bb_b + 1, int(e * len(bb) / 500) - 1))
        begins.append(bb_b)
        ends.append(bb_e)
    return dict(layers=layers, begins=begins, ends=ends, quals=quals if qual else None), truth


def run(n_windows=2000, threads=None, mode=0, seed=20260927, shape="ont"):
    rng = np.random.default_rng(seed)
    wins, truths = [], []
    for _ in range(n_windows):
        w, t = make_window(rng, shape=shape)
        wins.append(w)
        truths.append(t)
    eng = hip.Engine()
    eng.set_option("poa_rows_min_windows", 0)  # the chain of a full-size round (rows-on-lanes kernel first) whatever the batch size
    eng.poa_set_mode(mode)
    cons, status, ms = eng.poa_consensus_batch(wins)
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=threads or os.cpu_count() or 1) as ex:  # the oracle releases the GIL inside its C++ call
        refs = list(ex.map(lambda w: oracle.poa_window(w["layers"], begins=w["begins"], ends=w["ends"], quals=w["quals"])[0], wins))
    t_cpu = time.time() - t0
    same, diffs, unpolished = 0, [], 0
    for i, (c, r, st) in enumerate(zip(cons, refs, status)):
        if (int(st) & 0xFF) != 1:
            unpolished += 1
            continue
        if np.array_equal(c, r):
            same += 1
        else:
            # the same window by the oracle's statement of the device's rules (rows in the DEVICE's order, end node by smallest
            # id: only ties between equal scores can fall differently): identical then = the difference to spoa is such a tie
            r2 = oracle.poa_window(wins[i]["layers"], begins=wins[i]["begins"], ends=wins[i]["ends"], quals=wins[i]["quals"],
                                   device_order=True, end_tie=1)[0]
            d = oracle.edit_distance(bytes(c + 65), bytes(r + 65))
            dg = oracle.edit_distance(bytes(c + 65), bytes(truths[i] + 65))
            dr = oracle.edit_distance(bytes(r + 65), bytes(truths[i] + 65))
            diffs.append({"window": i, "layers": len(wins[i]["layers"]), "ed_device_vs_oracle": int(d), "ed_device_vs_truth": int(dg),
                          "ed_oracle_vs_truth": int(dr), "len_device": int(len(c)), "len_oracle": int(len(r)),
                          "identical_to_oracle_statement_of_device_rules": bool(np.array_equal(c, r2)), "status": int(st)})
    polished = n_windows - unpolished
    return {"windows": n_windows, "mode": mode, "shape": shape, "polished": polished, "identical": same,
            "identical_fraction": round(same / max(polished, 1), 6), "different": len(diffs), "device_ms": ms,
            "oracle_s": round(t_cpu, 1), "sum_ed_between": int(sum(x["ed_device_vs_oracle"] for x in diffs)),
            "device_closer_to_truth": int(sum(x["ed_device_vs_truth"] < x["ed_oracle_vs_truth"] for x in diffs)),
            "oracle_closer_to_truth": int(sum(x["ed_device_vs_truth"] > x["ed_oracle_vs_truth"] for x in diffs)),
            "equally_close": int(sum(x["ed_device_vs_truth"] == x["ed_oracle_vs_truth"] for x in diffs)),
            "different_explained_by_tie_rules": int(sum(x["identical_to_oracle_statement_of_device_rules"] for x in diffs)),
            "max_ed_between": int(max([x["ed_device_vs_oracle"] for x in diffs] or [0])),
            "not_explained": [x for x in diffs if not x["identical_to_oracle_statement_of_device_rules"]],
            "seed": seed, "examples": diffs[:12]}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    th = int(sys.argv[2]) if len(sys.argv) > 2 else None
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    shape = sys.argv[4] if len(sys.argv) > 4 else "ont"
    print(json.dumps(run(n, th, mode, shape=shape)))
