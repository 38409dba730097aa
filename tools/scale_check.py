#!/usr/bin/env python3
"""One-GPU run at a per-GPU shard size of BASELINE configs[3] (100 Mb / 8 GPUs = 12.5 Mb genome, 30x):
properties only (idempotence, CSR invariants, ground-truth precision).  python tools/scale_check.py [genome_bp]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raven_amd import hip, synth  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
t = time.time()
g = synth.make_genome(G, seed=99)
rs, truth = synth.make_reads(g, 30, 10000, length_model="lognormal", seed=98)
tg = time.time() - t
eng = hip.Engine()
rd = eng.upload(rs)
eng.set_timing(False)
res = []
for it in range(3):
    eng.reset_stats()
    t = time.time()
    p = eng.find_overlaps_and_create_piles(rd)
    dt = time.time() - t
    ovl, off = p.overlaps()
    data, poff = p.piles()
    res.append((dt, ovl, off, data))
    p.close()
assert all(np.array_equal(res[0][1], r[1]) and np.array_equal(res[0][3], r[3]) for r in res[1:]), "not idempotent"
dt, ovl, off, data = res[-1]
s, e = truth["start"], truth["start"] + truth["src_len"]
inter = np.minimum(e[ovl["lhs_id"]], e[ovl["rhs_id"]]) - np.maximum(s[ovl["lhs_id"]], s[ovl["rhs_id"]])
out = {"genome": G, "reads": rs.n, "bases": rs.total_bases, "max_read": int(rs.lengths.max()), "gen_s": round(tg, 1),
       "step_s": [round(r[0], 4) for r in res], "gbase_s": round(rs.total_bases / dt / 1e9, 3),
       "kept_overlaps": int(ovl.shape[0]), "true_overlap_fraction": float((inter > 0).mean()),
       "max_per_pile": int(np.diff(off.astype(np.int64)).max()), "mean_coverage": float(data.mean()),
       "counters": eng.counters()}
print(json.dumps(out))
