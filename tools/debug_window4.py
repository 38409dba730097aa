"""The window of tools/poa_parity.py's seeded set on which the device's branch completion lost a tie (hipcc 7.2 dropped the
predecessor update of the tie path in the lane-0 loop of poa.h): through the default chain, against the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import poa_parity as pp  # noqa: E402
from oracle import oracle  # noqa: E402
from raven_amd import hip  # noqa: E402

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 7327
rng = np.random.default_rng(20260927)
for _ in range(idx + 1):
    w, truth = pp.make_window(rng)
os.environ["RVN_POA4_MIN_WINDOWS"] = "0"
eng = hip.Engine()
ref = oracle.poa_window(w["layers"], begins=w["begins"], ends=w["ends"])[0]
for mode in (9, 2, 1):
    eng.poa_set_mode(mode)
    c, st, _ = eng.poa_consensus_batch([w])
    print("mode", mode, "len", len(c[0]), "oracle", len(ref), "equal", bool(np.array_equal(c[0], ref)))
