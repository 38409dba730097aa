"""One window of tools/poa_parity.py's seeded set through the device in different company (alone, in a slice, in the whole
batch, with one stream): does its consensus depend on the batch?  python tools/debug_window.py [index] [n_windows]"""
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
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    rng = np.random.default_rng(20260927)
    wins = [pp.make_window(rng)[0] for _ in range(n)]
    w = wins[idx]
    ref = oracle.poa_window(w["layers"], begins=w["begins"], ends=w["ends"])[0]
    eng = hip.Engine()

    def show(tag, cons, st):
        print(tag, "status", hex(int(st)), "len", len(cons), "equal to oracle", bool(np.array_equal(cons, ref)), flush=True)

    os.environ["RVN_POA4_MIN_WINDOWS"] = "0"
    for mode in (9, 2, 0):
        eng.poa_set_mode(mode)
        c, s, _ = eng.poa_consensus_batch([w])
        show("alone, mode %d" % mode, c[0], s[0])
    eng.poa_set_mode(9)
    lo = max(0, idx - 500)
    c, s, _ = eng.poa_consensus_batch(wins[lo:lo + 1000])
    show("slice of 1000, mode 9", c[idx - lo], s[idx - lo])
    for rep in range(2):
        c, s, _ = eng.poa_consensus_batch(wins)
        show("whole batch, mode 9, run %d" % rep, c[idx], s[idx])
    eng.poa_set_mode(0)
    c, s, _ = eng.poa_consensus_batch(wins)
    show("whole batch, mode 0", c[idx], s[idx])
    os.environ["RVN_POA4_STREAMS"] = "1"
    eng.poa_set_mode(9)
    c, s, _ = eng.poa_consensus_batch(wins)
    show("whole batch, mode 9, one stream", c[idx], s[idx])
    # its wave mates in the whole batch: the same four windows alone
    order = np.argsort([-len(x["layers"]) for x in wins], kind="stable")
    pos = int(np.where(order == idx)[0][0])
    print("position in longest-first order", pos, "layers", len(w["layers"]))


if __name__ == "__main__":
    main()
