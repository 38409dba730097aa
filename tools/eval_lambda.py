#!/usr/bin/env python3
"""Device polishing round on the lambda fixture vs the committed oracle result (tests/golden/lambda_polish.npz)."""
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from raven_amd import hip, seqio  # noqa: E402

golden = os.path.join(ROOT, "tests", "golden")
spec = importlib.util.spec_from_file_location("mg", os.path.join(golden, "make_golden_polish.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
rs, truth, draft, quals, avg_q = mg.inputs()
fx = np.load(os.path.join(golden, "lambda_polish.npz"))


def ed(a, b):
    return oracle.edit_distance(bytes(np.asarray(a, np.uint8) + 65), bytes(np.asarray(b, np.uint8) + 65))


out = {}
eng = hip.Engine(15, 5)
for mode in (0, 1):
    eng.poa_set_mode(mode)
    for wq in (True, False):
        cons, ratio, st = eng.polish_round(eng.upload(seqio.pack_reads([draft])), eng.upload(rs),
                                           quals=quals if wq else None, q=avg_q if wq else 0.0)
        ref = fx["consensus" if wq else "consensus_noqual"]
        out["mode%d_%s" % (mode, "qual" if wq else "noqual")] = dict(
            len_gpu=len(cons[0]), len_ref=len(ref), ed_gpu_ref=ed(cons[0], ref), ed_gpu_truth=ed(cons[0], truth),
            ed_ref_truth=ed(ref, truth), layers=st["n_layers"], dropped=st["n_dropped_layers"], used=st["n_reads_used"],
            wide=eng.poa_wide_windows(), fallback=eng.poa_fallback_windows())
print(json.dumps(out))
