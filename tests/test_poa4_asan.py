"""Memory safety of the window-consensus kernel source (raven_amd/csrc/poa4.hip) without a GPU: the kernel is written
against sv:: (csrc/simt.h) and runs under the host wavefront emulator, so its HOST compilation can be instrumented with
AddressSanitizer (ROCm's clang ships the runtime).  Every access of the phase functions is then checked against the
allocation it falls into: the wave's LDS image (its own allocation: an index beyond the 9 KB the kernel declares is
caught), the windows' state records, the batch description, the layers' codes and qualities, the output, and the scratch of
the batch (ONE allocation for all its window slots: an access beyond the batch's scratch is caught, one that strays into a
neighbouring field of a slot is not) — on windows that exercise partial layers, qualities, ragged groups and the limits.
UndefinedBehaviorSanitizer rides along: a shift by 32 or more, a signed overflow, a misaligned access are where the C++ the
emulator executes and the instructions the GPU executes could part ways without either being "wrong".  Results are compared with the
oracle as in tests/test_poa4_emulation.py (the instrumented build must not change them).

The instrumented library is built into the test's temporary directory from poa4.hip / poa.hip / simt_emu.hip + the test
library's other objects; nothing under raven_amd/lib is touched."""
import glob
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "raven_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
UNITS = ["scan", "radix_sort", "sketch", "index", "map", "pile", "edit_distance", "poa", "poa2", "poa4", "polish", "nwpath",
         "pass2", "io", "shard", "group", "engine", "edlib_dropin", "simt_emu"]
INSTRUMENTED = ["poa4", "poa", "simt_emu"]

SCRIPT = r"""
import numpy as np
from oracle import oracle
from raven_amd import hip
from tests.test_poa4_emulation import _window, _oracle
rng = np.random.default_rng(31)
wins = [_window(rng, int(rng.integers(40, 300)), int(rng.integers(3, 16)), partial=0.3 if i % 2 else 0.0, qual=(i % 3 == 0)) for i in range(14)]
wins += [_window(rng, n, 5, err=(0.03, 0.02, 0.02)) for n in (5, 17, 33)]
bb = rng.integers(0, 4, size=950, dtype=np.uint8)
wins += [dict(layers=[bb, bb.copy(), bb.copy()])]              # beyond the kernel's length limit: reported, not touched
wins += [_window(rng, 500, 30, partial=0.25)]                   # the shape a polishing round produces
for variant in (5, 4):
    cons, status = hip.poa_banded_emulate(wins, variant=variant)
    polished = 0
    for w, c, st in zip(wins, cons, status):
        if (int(st) & 0xFF) == 1:
            polished += 1
            assert np.array_equal(c, _oracle(w)), variant
    assert polished >= 14, (variant, status)
    print("variant", variant, "polished", polished, "of", len(wins))
"""


def _asan_runtime():
    for p in glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"):
        return p
    return None


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_window_consensus_kernel_source_under_address_and_ub_sanitizers(tmp_path):
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("this ROCm has no shared AddressSanitizer runtime")
    others = [os.path.join(CSRC, "obj_test", u + ".o") for u in UNITS if u not in INSTRUMENTED]
    if not all(os.path.exists(o) for o in others):
        pytest.skip("the test library's objects are not in the tree (raven_amd/csrc/build.sh leaves them in obj_test/)")
    flags = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-Wno-unused-function", "-DRVN_TEST_HOOKS", "-DRVN_DEBUG_KNOBS",
             "-fsanitize=address,undefined", "-fno-sanitize=vptr,function", "-fno-gpu-sanitize", "-shared-libsan"]
    jobs = [subprocess.Popen([HIPCC] + flags + ["-c", os.path.join(CSRC, u + ".hip"), "-o", str(tmp_path / (u + ".o"))],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for u in INSTRUMENTED]
    for u, j in zip(INSTRUMENTED, jobs):
        _, err = j.communicate(timeout=1500)
        assert j.returncode == 0, (u, err[-2000:])
    lib = str(tmp_path / "libraven_hip_test.so")
    objs = [str(tmp_path / (u + ".o")) if u in INSTRUMENTED else os.path.join(CSRC, "obj_test", u + ".o") for u in UNITS]
    link = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan",
                           "-o", lib] + objs + ["-lz"], capture_output=True, text=True, timeout=900)
    assert link.returncode == 0, link.stderr[-2000:]
    env = dict(os.environ, RVN_LIB_PATH=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", PYTHONPATH=ROOT)
    run = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert run.returncode == 0 and "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, (run.stdout[-1000:], run.stderr[-3000:])
    assert run.stdout.count("polished") == 2, run.stdout
