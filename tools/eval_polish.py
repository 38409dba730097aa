#!/usr/bin/env python3
"""Accuracy of rvn_polish_round against the CPU restatement of racon's round (exact CIGAR breakpoints) on small
cases: edit distance to the truth of draft / device / CPU, with and without trimming."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from raven_amd import hip, seqio, synth  # noqa: E402


def ed(a, b):
    return oracle.edit_distance(bytes(np.asarray(a, np.uint8) + 65), bytes(np.asarray(b, np.uint8) + 65))


def main():
    out = []
    eng = hip.Engine(15, 5)
    for glen, cov, rlen, seed in ((12_000, 20, 2000, 7), (24_000, 25, 2500, 17), (20_000, 30, 8000, 27)):
        g = synth.make_genome(glen, seed=seed)
        draft = synth.make_draft(g, seed=seed + 1)
        reads, _ = synth.make_reads(g, cov, rlen, seed=seed + 2)
        targets = seqio.pack_reads([draft])
        row = {"genome": glen, "cov": cov, "read_len": rlen, "ed_draft": ed(draft, g)}
        for trim in (False, True):
            cons, ratio, st = eng.polish_round(eng.upload(targets), eng.upload(reads), trim=trim)
            ref, _ = oracle.polish_round(targets, reads, trim=trim)
            row["trim" if trim else "notrim"] = {"ed_gpu": ed(cons[0], g), "ed_cpu": ed(ref[0], g), "ed_gpu_cpu": ed(cons[0], ref[0]),
                                                 "len_gpu": len(cons[0]), "len_cpu": len(ref[0]), "dropped": st["n_dropped_layers"]}
        out.append(row)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
