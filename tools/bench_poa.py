#!/usr/bin/env python3
"""Throughput + agreement of rvn_poa_consensus_batch on racon-like windows (500 bp backbone, ~30 ONT-like layers).
    python tools/bench_poa.py [n_windows] [n_check]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raven_amd import hip  # noqa: E402


def mutate_many(rng, truth, n, sub, ins, dele):
    out = []
    L = truth.shape[0]
    for _ in range(n):
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
        out.append(seq.astype(np.uint8))
    return out


def main():
    n_windows = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    n_check = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(11)
    wins, truths = [], []
    cells = 0
    for _ in range(n_windows):
        truth = rng.integers(0, 4, size=500, dtype=np.uint8)
        bb = mutate_many(rng, truth, 1, 0.03, 0.02, 0.02)[0]
        reads = mutate_many(rng, truth, 30, 0.04, 0.03, 0.03)
        wins.append(dict(layers=[bb] + reads, quals=None))
        truths.append(truth)
        cells += sum(len(r) for r in reads) * 650  # ~nodes x layer length, order of magnitude
    eng = hip.Engine()
    # RVN_POA_MODES="9,2": the same windows through several kernels in one process (one JSON line each)
    # an entry may carry environment overrides for that run: "9@RVN_POA_WAVES_PER_CU=6@RVN_POA_NMAX_MULT=3"
    for entry in os.environ.get("RVN_POA_MODES", os.environ.get("RVN_POA_MODE", "0")).split(","):
        parts = entry.split("@")
        saved = {}
        for kv in parts[1:]:
            k, v = kv.split("=")
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        # RVN_POA_MIN=n: the engine option poa_rows_min_windows for this run (0: the first attempt whatever the batch size)
        eng.set_option("poa_rows_min_windows", int(os.environ.get("RVN_POA_MIN", "-1")))
        run_mode(eng, int(parts[0]), wins, truths, n_windows, n_check, cells, entry)
        for k, v in saved.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def run_mode(eng, mode, wins, truths, n_windows, n_check, cells, label=""):
    eng.poa_set_mode(mode)
    eng.poa_consensus_batch(wins[:64])  # warm-up / allocation
    ms = None
    for _ in range(int(os.environ.get("RVN_POA_REPS", "1"))):  # (the first full-size call also grows the per-wave scratch)
        t = time.time()
        cons, status, ms1 = eng.poa_consensus_batch(wins)
        wall = time.time() - t
        ms = ms1 if ms is None else min(ms, ms1)
    out = {"mode": mode, "run": label, "windows": n_windows, "layers_per_window": 30, "device_ms": ms, "wall_s": wall,
           "windows_per_s": n_windows / ms * 1e3, "approx_gcups": cells / ms / 1e6,
           "status_counts": {int(k): int(v) for k, v in zip(*np.unique(status & 0xFF, return_counts=True))},
           "fail_layers": [(int(x) >> 8) & 0xFFFF for x in status[status > 1][:20]], "fail_windows": [int(i) for i in np.nonzero(status > 1)[0][:20]],
           "phase_cycles": eng.poa_phase_cycles(), "fallback_windows": eng.poa_fallback_windows(), "wide_windows": eng.poa_wide_windows(),
           "kernel_ms": {k: v for k, v in eng.kernel_ms().items() if k.startswith("poa")} if hasattr(eng, "kernel_ms") else None,
           "read_bases_per_s": sum(sum(len(x) for x in w["layers"][1:]) for w in wins) / ms * 1e3}
    if n_check:
        from oracle import oracle
        same, ed_sum, ed_truth_gpu, ed_truth_cpu = 0, 0, 0, 0
        t = time.time()
        for w, c, tr in list(zip(wins, cons, truths))[:n_check]:
            ref, _ = oracle.poa_window(w["layers"])
            d = oracle.edit_distance(bytes(c + 65), bytes(ref + 65))
            same += d == 0
            ed_sum += d
            ed_truth_gpu += oracle.edit_distance(bytes(c + 65), bytes(tr + 65))
            ed_truth_cpu += oracle.edit_distance(bytes(ref + 65), bytes(tr + 65))
        out.update({"checked": n_check, "identical_to_cpu": same, "sum_ed_gpu_vs_cpu": ed_sum,
                    "sum_ed_gpu_vs_truth": ed_truth_gpu, "sum_ed_cpu_vs_truth": ed_truth_cpu,
                    "cpu_oracle_s_per_window": (time.time() - t) / n_check})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
