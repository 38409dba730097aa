"""GPU check of the four-windows-per-wave POA kernel (poa3.hip, mode 5) against the one-window-per-wave kernel
(poa2.hip, mode 2) on synthetic windows: statuses and consensus must be identical window for window; prints timings."""
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
    for mode in (2, 5, 2, 5):
        eng.poa_set_mode(mode)
        t = time.time()
        cons, status, ms = eng.poa_consensus_batch(wins)
        res[mode] = (cons, status.copy())
        print("mode", mode, "device_ms %.1f wall %.2f" % (ms, time.time() - t), "status histogram",
              dict(zip(*np.unique(status & 0xFF, return_counts=True))), flush=True)
    c2, s2 = res[2]
    c5, s5 = res[5]
    bad = 0
    for i in range(n):
        if (s2[i] & 0xFF) == 1 and (s5[i] & 0xFF) == 1:
            if not np.array_equal(c2[i], c5[i]):
                bad += 1
                if bad < 10:
                    print("window", i, "differs", len(c2[i]), len(c5[i]))
        elif s2[i] != s5[i]:
            print("window", i, "status", s2[i], s5[i])
    print("compared", n, "different", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
