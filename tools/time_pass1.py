#!/usr/bin/env python3
"""Wall time, per-stage device time (HIP events with a sync per stage) and per-kernel time of the first pass:
    python tools/time_pass1.py [genome_bases] [ont|ont10k]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raven_amd import hip, synth  # noqa: E402

genome = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
model = sys.argv[2] if len(sys.argv) > 2 else "ont10k"
dev = torch.device("cuda", 0)
g = synth.make_genome_torch(genome, seed=11, device=dev)
if model == "ont":
    rs, _ = synth.make_reads_torch(g, 30, 9000, length_model="lognormal", sub=0.04, ins=0.03, dele=0.03, seed=12)
else:
    rs, _ = synth.make_reads_torch(g, 30, 10000, length_model="fixed", sub=0.04, ins=0.03, dele=0.03, seed=12)
eng = hip.Engine(15, 5)
rd = eng.upload(rs)
for stage_timing in (False, True, False):
    eng.set_timing(stage_timing)
    eng.set_kernel_timing(True)
    for i in range(2):
        eng.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p = eng.find_overlaps_and_create_piles(rd)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        p.close()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        km = eng.kernel_ms()
        print("stage_timing=%s pass %d: wall %.1f ms (+ close %.1f ms), kernels %.1f ms" %
              (stage_timing, i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, sum(v[0] for v in km.values())))
        if stage_timing:
            print("   stages:", {k: round(v[0], 1) for k, v in eng.stage_ms().items() if v[0] > 0.5})
