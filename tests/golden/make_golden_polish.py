#!/usr/bin/env python3
"""Generates tests/golden/lambda_polish.npz: one racon-style polishing round, computed by the CPU oracle
(oracle.polish_round = restatement of racon's pipeline: whole-overlap NW path -> CIGAR breakpoints -> windows -> spoa
consensus), on the reference's own test data: target = NC_001416 (lambda) with seeded draft errors, reads =
ERA476754 with the block-mean qualities Raven hands to racon (biosoup block_quality: mean Phred per 64 bases) and
Raven's threshold q = average of the reads' mean block quality (RavenLib/src/polish.cc:25-47).

Like lambda_pass1.npz these vectors come from the oracle restatement, NOT from racon itself (racon/spoa/edlib are
absent): they pin oracle and HIP path against each other and against regressions.
    python tests/golden/make_golden_polish.py        (about 2 minutes of CPU)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle  # noqa: E402
from raven_amd import seqio, synth  # noqa: E402
from oracle import seqio_oracle


def block_qualities(rs):
    """Per-base Phred+33 as Raven/racon see it: the mean of each 64-base block (biosoup), and Raven's avg_q."""
    per_base, means = [], []
    for i in range(rs.n):
        q = np.asarray(rs.qualities[i], dtype=np.float64)
        nb = (q.shape[0] + 63) // 64
        blocks = np.array([int(q[b * 64:(b + 1) * 64].mean()) for b in range(nb)], dtype=np.uint8)
        means.append(blocks.astype(np.float64).mean())
        per_base.append((np.repeat(blocks, 64)[:q.shape[0]] + 33).astype(np.uint8))
    return per_base, float(np.mean(means))


def inputs():
    rs = seqio_oracle.load_reads(os.path.join(HERE, "ERA476754.fastq.gz"))
    ref = seqio_oracle.load_reads(os.path.join(HERE, "NC_001416.fasta.gz"))
    truth = ref.codes(0)
    draft = synth.make_draft(truth, seed=20260926)
    quals, avg_q = block_qualities(rs)
    return rs, truth, draft, quals, avg_q


def main():
    rs, truth, draft, quals, avg_q = inputs()
    targets = seqio.pack_reads([draft])
    cons, ratio = oracle.polish_round(targets, rs, quals=quals, q=avg_q)
    cons_nq, ratio_nq = oracle.polish_round(targets, rs)
    # the integer half of the round (mapping, best overlap, alignment path, breakpoints, layer rules): bit-exact target
    layers = oracle.polish_layers(targets, rs, quals=quals, q=avg_q)
    layers_nq = oracle.polish_layers(targets, rs)
    ed = lambda a: oracle.edit_distance(bytes(np.asarray(a, np.uint8) + 65), bytes(truth + 65))  # noqa: E731
    out = dict(draft=draft, avg_q=np.array([avg_q]), consensus=cons[0], ratio=np.array([ratio[0]]),
               consensus_noqual=cons_nq[0], ratio_noqual=np.array([ratio_nq[0]]), layers=layers, layers_noqual=layers_nq,
               ed_draft=np.array([ed(draft)]), ed_consensus=np.array([ed(cons[0])]), ed_consensus_noqual=np.array([ed(cons_nq[0])]))
    np.savez_compressed(os.path.join(HERE, "lambda_polish.npz"), **out)
    print("wrote lambda_polish.npz", {k: (v.shape, v[:1]) for k, v in out.items()})


if __name__ == "__main__":
    main()
