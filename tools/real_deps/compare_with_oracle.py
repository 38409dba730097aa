#!/usr/bin/env python3
"""Diffs the real ram / edlib / racon (oracle/_ref/ref_harness, built by tools/fetch_real_deps.sh) with oracle/ on the
reference's lambda data.  Exit status 0 = every comparison identical (the oracle can then be called pinned)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from raven_amd import seqio  # noqa: E402
from oracle import seqio_oracle

READS = os.path.join(ROOT, "tests", "golden", "ERA476754.fastq.gz")
GENOME = os.path.join(ROOT, "tests", "golden", "NC_001416.fasta.gz")


def run(harness, *args):
    return subprocess.run([harness, *map(str, args)], check=True, capture_output=True, text=True).stdout.splitlines()


def main():
    harness = sys.argv[1]
    rs = seqio_oracle.load_reads(READS)
    bad = 0
    for minhash in (0, 1):
        real = np.array([[int(x) for x in ln.split()] for ln in run(harness, "map", READS, 15, 5, 0.001, minhash)], dtype=np.int64)
        eng = oracle.Engine(15, 5)
        eng.minimize(rs, 0, rs.n, bool(minhash))
        eng.filter(0.001)
        mine = []
        for i in range(rs.n):
            for o in eng.map(rs, i, True, True, bool(minhash))["overlaps"]:
                mine.append([int(o[f]) for f in ("lhs_id", "lhs_begin", "lhs_end", "rhs_id", "rhs_begin", "rhs_end", "score", "strand")])
        mine = np.array(mine, dtype=np.int64).reshape(-1, 8)
        same = real.shape == mine.shape and np.array_equal(real, mine)
        print("ram Map minhash=%d: real %d overlaps, oracle %d -> %s" % (minhash, real.shape[0], mine.shape[0], "IDENTICAL" if same else "DIFFERENT"))
        bad += not same
    d_real = [int(x) for x in run(harness, "edlib", READS, 40)]
    d_mine = [oracle.edit_distance(rs.inflate(i), rs.inflate(i + 1)) for i in range(len(d_real))]
    print("edlib distances:", "IDENTICAL" if d_real == d_mine else "DIFFERENT")
    bad += d_real != d_mine
    cons_real = run(harness, "polish", READS, GENOME)
    targets = seqio_oracle.load_reads(GENOME)
    cons_mine, _ = oracle.polish_round(targets, rs)[:2]
    ok = len(cons_real) == len(cons_mine) and all(a == "".join("ACGT"[c] for c in b) for a, b in zip(cons_real, cons_mine))
    print("racon round on lambda:", "IDENTICAL" if ok else "DIFFERENT")
    bad += not ok
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
