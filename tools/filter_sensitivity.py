#!/usr/bin/env python3
"""What the two recollected details of ram that SURVEY.md Appendix A flags as uncertain are worth on a data set
(default: the reference's own lambda reads, tests/golden/ERA476754.fastq.gz):
  * Filter's "+ 1" (occurrence_ = count at the (1 - f) quantile + 1; Map skips keys with count > occurrence_),
  * minhash's resize(len / k).
Runs the CPU oracle's Map over every read with occurrence - 1 / occurrence / occurrence + 1 and reports how many
overlaps and how many reads' overlap lists change.  It pins nothing (the oracle stays "parity unpinned" for ram); it
says how far a wrong recollection could move the result:  python tools/filter_sensitivity.py [reads.fastq.gz] [freq]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from raven_amd import seqio  # noqa: E402
from oracle import seqio_oracle


def map_all(eng, rs, minhash):
    out = []
    for i in range(rs.n):
        o = eng.map(rs, i, True, True, minhash)["overlaps"]
        out.append(o.tobytes())
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ERA476754.fastq.gz")
    freq = float(sys.argv[2]) if len(sys.argv) > 2 else 0.001
    rs = seqio_oracle.load_reads(path)
    report = {"reads": rs.n, "bases": int(rs.total_bases), "freq": freq}
    for minhash in (False, True):
        eng = oracle.Engine(15, 5)
        eng.minimize(rs, 0, rs.n, minhash)
        eng.filter(freq)
        occ = eng.occurrence
        base = map_all(eng, rs, minhash)
        row = {"occurrence": occ, "overlaps": sum(len(b) for b in base) // 32}
        for delta in (-1, +1):
            eng.set_occurrence(max(occ + delta, 0))
            alt = map_all(eng, rs, minhash)
            row["occurrence%+d" % delta] = {
                "overlaps": sum(len(b) for b in alt) // 32,
                "reads_with_a_different_list": int(sum(a != b for a, b in zip(alt, base)))}
        report["minhash" if minhash else "all_minimizers"] = row
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
