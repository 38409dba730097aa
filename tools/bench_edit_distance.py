#!/usr/bin/env python3
"""Throughput of rvn_edit_distance_batch on overlap-like span pairs (GCUPS = n*m DP cells / device time,
i.e. what edlib would have to cover without a band).  python tools/bench_edit_distance.py [pairs] [len]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raven_amd import hip, synth  # noqa: E402


def run(tag, n_pairs, length, sub, ins, dele):
    g = synth.make_genome(max(4 * length, 200_000), seed=5)
    # two independent noisy copies of the same locus per pair = what the identity filter sees
    rs_a, tr_a = synth.make_reads(g, n_pairs * length / g.shape[0], length, seed=6, sub=sub, ins=ins, dele=dele)
    n = rs_a.n // 2
    eng = hip.Engine()
    rd = eng.upload(rs_a)
    pairs = np.zeros(0, dtype=hip.ED_PAIR_DTYPE)
    # pair reads that overlap on the genome: sort by start and pair neighbours, restricted to the shared span
    order = np.argsort(tr_a["start"])
    P = []
    for x, y in zip(order[:-1], order[1:]):
        sx, sy = int(tr_a["start"][x]), int(tr_a["start"][y])
        shared = sx + length - sy
        if shared < 1000 or tr_a["strand"][x] != 0 or tr_a["strand"][y] != 0:
            continue
        la, lb = int(rs_a.lengths[x]), int(rs_a.lengths[y])
        a_begin = min(la - 1, int((sy - sx) * la / length))
        P.append((x, a_begin, la - a_begin, y, 0, min(lb, int(shared * lb / length)), 1, 0))
        if len(P) >= n_pairs:
            break
    pairs = np.array(P, dtype=hip.ED_PAIR_DTYPE)
    eng.edit_distance_batch(rd, pairs[:8])  # warm-up
    d, ms, cells = eng.edit_distance_batch(rd, pairs)
    ident = 1.0 - d / np.maximum(pairs["lhs_len"], pairs["rhs_len"])
    return {"case": tag, "pairs": int(pairs.shape[0]), "mean_len": float(pairs["lhs_len"].mean()),
            "mean_edit_distance": float(d.mean()), "mean_identity": float(ident.mean()), "device_ms": ms,
            "dp_cells": int(cells), "gcups_nm": cells / ms / 1e6, "pairs_per_s": pairs.shape[0] / ms * 1e3}


if __name__ == "__main__":
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    length = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    out = [run("hifi-like 0.5% error", n_pairs, length, 0.001, 0.002, 0.002),
           run("ont-like 10% error", n_pairs, length, 0.04, 0.03, 0.03)]
    print(json.dumps(out))
