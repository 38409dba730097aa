"""Where a device consensus differs from the oracle's (one window of tools/poa_parity.py's seeded set), with and without
racon's coverage trim, and after how many layers the two part ways.  python tools/debug_window2.py [index]"""
import difflib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import poa_parity as pp  # noqa: E402
from oracle import oracle  # noqa: E402
from raven_amd import hip  # noqa: E402


def s(x):
    return "".join("ACGT"[int(v)] for v in x)


def main():
    idx = int(sys.argv[1]) if len(sys.argv) > 1 else 7327
    rng = np.random.default_rng(20260927)
    w = None
    for _ in range(idx + 1):
        w, truth = pp.make_window(rng)
    eng = hip.Engine()
    os.environ["RVN_POA4_MIN_WINDOWS"] = "0"
    eng.poa_set_mode(9)
    for trim in (True, False):
        ref = oracle.poa_window(w["layers"], begins=w["begins"], ends=w["ends"], trim=trim)[0]
        c, st, _ = eng.poa_consensus_batch([w], trim=trim)
        ops = [o for o in difflib.SequenceMatcher(None, s(c[0]), s(ref), autojunk=False).get_opcodes() if o[0] != "equal"]
        print("trim", trim, "device", len(c[0]), "oracle", len(ref), "ops", ops, flush=True)
    # first layer count at which the two differ (no trim)
    k_first = None
    subs = []
    for k in range(3, len(w["layers"]) + 1):
        subs.append(dict(layers=w["layers"][:k], begins=w["begins"][:k], ends=w["ends"][:k], quals=None))
    cs, sts, _ = eng.poa_consensus_batch(subs, trim=False)
    for k, c in zip(range(3, len(w["layers"]) + 1), cs):
        ref = oracle.poa_window(w["layers"][:k], begins=w["begins"][:k], ends=w["ends"][:k], trim=False)[0]
        if not np.array_equal(c, ref):
            k_first = k
            ops = [o for o in difflib.SequenceMatcher(None, s(c), s(ref), autojunk=False).get_opcodes() if o[0] != "equal"]
            print("first difference with", k, "layers (the last one spans", w["begins"][k - 1], w["ends"][k - 1], "len", len(w["layers"][k - 1]), "):", ops)
            break
    print("k_first", k_first, "of", len(w["layers"]))


if __name__ == "__main__":
    main()
