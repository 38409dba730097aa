import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from raven_amd import hip
from test_gpu_poa import _window
rng = np.random.default_rng(7)
wins = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    w, _ = _window(rng, length=int(rng.integers(300, 560)), n_reads=int(rng.integers(5, 34)), err=(0.05, 0.04, 0.04), partial=0.25 if i % 2 else 0.0, qual=(i % 3 == 0))
    wins.append(w)
eng = hip.Engine()
eng.poa_set_mode(2)
c2, s2, _ = eng.poa_consensus_batch(wins)
print("mode 2 ok", np.unique(s2 & 0xFF, return_counts=True), flush=True)
eng.poa_set_mode(0)
eng.set_option("poa_rows_min_windows", 0)
c0, s0, _ = eng.poa_consensus_batch(wins)
print("mode 0 ok", np.unique(s0 & 0xFF, return_counts=True), "to64", eng.poa_narrow_windows() if hasattr(eng, "poa_narrow_windows") else None, flush=True)
print("status equal", np.array_equal(s0 & 0xFF, s2 & 0xFF), "cons equal", all(np.array_equal(a, b) for a, b in zip(c0, c2)))
