#!/usr/bin/env python3
"""One racon-style polishing round on the device (rvn_polish_round) on a synthetic draft:
    python tools/bench_polish.py [genome_len] [coverage] [read_len] [rounds]
Reports read Gbase/s through the round and where the time goes (map / host / POA kernel)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raven_amd import hip, seqio, synth  # noqa: E402
from tests import polish_util  # noqa: E402


def main():
    glen = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    cov = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rlen = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    rng = np.random.default_rng(3)
    g = synth.make_genome(glen, seed=103)
    draft = polish_util.mutate(rng, g, 0.01, 0.008, 0.008)
    reads, _ = synth.make_reads(g, cov, rlen, seed=203)
    eng = hip.Engine(15, 5)
    eng.poa_set_mode(int(os.environ.get("RVN_POA_MODE", "0")))
    rd = eng.upload(reads)
    cur = seqio.pack_reads([draft])
    out = {"genome": glen, "coverage": cov, "read_len": rlen, "read_bases": int(reads.lengths.sum()), "rounds": []}
    eng.polish_round(eng.upload(seqio.pack_reads([draft[:50_000]])), rd)  # warm-up (allocations)
    for r in range(rounds):
        t = time.time()
        cons, ratio, st = eng.polish_round(eng.upload(cur), rd)
        wall = time.time() - t
        st.update({"wall_s": wall, "ratio": float(ratio[0]), "len": int(len(cons[0])),
                   "read_gbase_per_s": out["read_bases"] / wall / 1e9, "fallback": eng.poa_fallback_windows(),
                   "wide": eng.poa_wide_windows()})
        out["rounds"].append(st)
        cur = seqio.pack_reads([cons[0]])
    # accuracy on a 20 kb prefix (CPU checker is quadratic; end effects are a few bases)
    from oracle import oracle
    a = bytes(g[:20_000] + 65)
    out["prefix_ed_draft_vs_truth"] = oracle.edit_distance(bytes(draft[:20_000] + 65), a)
    out["prefix_ed_polished_vs_truth"] = oracle.edit_distance(bytes(cons[0][:20_000] + 65), a)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
