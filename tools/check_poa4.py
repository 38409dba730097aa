"""GPU check of the rows-on-lanes POA kernel (poa4.hip, mode 9) against the one-row-per-iteration kernel (poa2.hip,
mode 2) on synthetic windows: consensus must be identical window for window wherever both polish; prints timings.
usage: check_poa4.py [n_windows] [modes, e.g. 9,0]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from raven_amd import hip  # noqa: E402
from test_gpu_poa import _window  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rng = np.random.default_rng(7)
    wins = []
    for i in range(n):
        w, _ = _window(rng, length=int(rng.integers(300, 560)), n_reads=int(rng.integers(5, 34)), err=(0.05, 0.04, 0.04),
                       partial=0.25 if i % 2 else 0.0, qual=(i % 3 == 0))
        wins.append(w)
    eng = hip.Engine()
    res = {}
    modes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [9]
    for mode in [2] + modes:
        eng.poa_set_mode(mode)
        t = time.time()
        cons, status, ms = eng.poa_consensus_batch(wins)
        res[mode] = (cons, status.copy())
        print("mode", mode, "device_ms %.1f wall %.2f" % (ms, time.time() - t), "status histogram",
              dict(zip(*np.unique(status & 0xFF, return_counts=True))), flush=True)
    c2, s2 = res[2]
    total_bad = 0
    for mode in modes:  # mode 9 uses a 32-column band: more windows flagged (8), the polished ones must agree
        c5, s5 = res[mode]
        bad = both = 0
        for i in range(n):
            if (s2[i] & 0xFF) == 1 and (s5[i] & 0xFF) == 1:
                both += 1
                if not np.array_equal(c2[i], c5[i]):
                    bad += 1
                    if bad < 10:
                        print("mode", mode, "window", i, "differs", len(c2[i]), len(c5[i]))
        print("mode", mode, "polished by both", both, "of", n, "different", bad)
        total_bad += bad
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
