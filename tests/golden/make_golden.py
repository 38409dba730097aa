#!/usr/bin/env python3
"""Generates tests/golden/lambda_pass1.npz: per-stage dumps of the CPU oracle on the reference's own test
reads (RavenTest/data/ERA476754.fastq.gz, copied here as a data fixture).

The reference itself cannot be built in this container, so these vectors come from the oracle
restatement, NOT from lbcb-sci/raven: they pin the oracle and the HIP path against regressions and against
each other; they do not pin either against the real ram.   python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle  # noqa: E402
from raven_amd import seqio  # noqa: E402
from oracle import seqio_oracle


def main():
    rs = seqio_oracle.load_reads(os.path.join(HERE, "ERA476754.fastq.gz"))
    out = {"n_reads": np.array([rs.n]), "total_bases": np.array([rs.total_bases])}
    e = oracle.Engine(15, 5)
    for i in range(4):
        for mh in (0, 1):
            v, o = e.sketch(rs, i, bool(mh))
            out["sketch_%d_%d_values" % (i, mh)] = v
            out["sketch_%d_%d_origins" % (i, mh)] = o
    for mh in (0, 1):
        r = oracle.Engine(15, 5).find_overlaps_and_create_piles(rs, freq=0.001, kmax=32, use_minhash=bool(mh))
        out["pass1_%d_occurrence" % mh] = np.array([r["occurrence"]], dtype=np.uint32)
        out["pass1_%d_overlaps" % mh] = r["overlaps"]
        out["pass1_%d_overlap_offsets" % mh] = r["overlap_offsets"]
        out["pass1_%d_pile_data" % mh] = r["pile_data"]
        out["pass1_%d_pile_offsets" % mh] = r["pile_offsets"]
        c = r["counters"]
        out["pass1_%d_counters" % mh] = np.array([c[k] for k in sorted(c)], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "lambda_pass1.npz"), **out)
    print("wrote lambda_pass1.npz", {k: v.shape for k, v in out.items() if "pass1_1" in k})


if __name__ == "__main__":
    main()
