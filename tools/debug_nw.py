#!/usr/bin/env python3
"""Two polishing rounds on a small synthetic set with the alignment stage's debug output (RVN_NW_DEBUG=2)."""
import faulthandler
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raven_amd import hip, seqio, synth  # noqa: E402
from tests import polish_util  # noqa: E402

faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "60")), exit=True)
glen = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
rlen = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
rng = np.random.default_rng(3)
g = synth.make_genome(glen, seed=103)
draft = polish_util.mutate(rng, g, 0.01, 0.008, 0.008)
reads, _ = synth.make_reads(g, 30, rlen, seed=203)
eng = hip.Engine(15, 5)
rd = eng.upload(reads)
cur = seqio.pack_reads([draft])
for r in range(3):
    t = time.time()
    cons, ratio, st = eng.polish_round(eng.upload(cur), rd)
    print("round", r, "wall %.3f" % (time.time() - t), {k: st[k] for k in ("align_ms", "poa_ms", "n_aligned", "n_align_retries")}, flush=True)
