"""One window of tools/poa_parity.py's seeded set through every kernel of the chain, and through the rows-on-lanes kernel
with other scratch geometries: python tools/debug_window3.py [index]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import poa_parity as pp  # noqa: E402
from oracle import oracle  # noqa: E402
from raven_amd import hip  # noqa: E402


def main():
    idx = int(sys.argv[1]) if len(sys.argv) > 1 else 7327
    rng = np.random.default_rng(20260927)
    for _ in range(idx + 1):
        w, truth = pp.make_window(rng)
    os.environ["RVN_POA4_MIN_WINDOWS"] = "0"
    ref = oracle.poa_window(w["layers"], begins=w["begins"], ends=w["ends"], trim=False)[0]
    eng = hip.Engine()
    for mode in (1, 2, 3, 4, 9):
        eng.poa_set_mode(mode)
        c, st, _ = eng.poa_consensus_batch([w], trim=False)
        print("mode", mode, "status", hex(int(st[0])), "len", len(c[0]), "oracle", len(ref), "equal", bool(np.array_equal(c[0], ref)), flush=True)
    eng.poa_set_mode(9)
    for env in ({"RVN_POA_NMAX_MULT": "3"}, {"RVN_POA_NMAX_MULT": "12"}, {"RVN_POA4_UPD": "2"}):
        os.environ.update(env)
        c, st, _ = eng.poa_consensus_batch([w], trim=False)
        print(env, "status", hex(int(st[0])), "len", len(c[0]), "equal", bool(np.array_equal(c[0], ref)), flush=True)
        for k in env:
            del os.environ[k]
    # the same layers in another order of the full-span ones / without the qualities-free weights: weights 2 everywhere
    q = [np.full(len(x), 35, dtype=np.uint8) for x in w["layers"]]
    c, st, _ = eng.poa_consensus_batch([dict(w, quals=q)], trim=False)
    refq = oracle.poa_window(w["layers"], begins=w["begins"], ends=w["ends"], quals=q, trim=False)[0]
    print("weights 2", "len", len(c[0]), "oracle", len(refq), "equal", bool(np.array_equal(c[0], refq)))


if __name__ == "__main__":
    main()
