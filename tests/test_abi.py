"""The C-ABI library loads and exports every symbol include/raven_hip.h declares; with no GPU present the
product path fails loudly instead of falling back to anything."""
import os
import re

import pytest

from raven_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="raven_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rvn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    L = hip.lib()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "missing export: " + name
    assert sorted(hip.SYMBOLS) == declared, "raven_amd/hip.py SYMBOLS out of sync with include/raven_hip.h"


def test_hooks_live_in_the_test_library_only():
    """a drop-in does not export an emulator.  libraven_hip.so exports NO rvn_test_* symbol, no
    rvn_poa_banded_emulate and nothing of simt_emu; libraven_hip_test.so (include/raven_hip_test.h) exports the hooks on top
    of everything the product header declares."""
    import subprocess
    hooks = _declared_symbols("raven_hip_test.h")
    assert sorted(hip.TEST_SYMBOLS) == hooks and len(hooks) == 12
    names = subprocess.check_output(["nm", "-D", "--defined-only", hip.LIB_PATH]).decode()
    assert "rvn_test_" not in names and "emulate" not in names and "simt_emu" not in names
    T = hip.test_lib()
    for name in hooks + _declared_symbols():
        assert hasattr(T, name), "missing export in libraven_hip_test.so: " + name


def test_edlib_dropin_symbols_exported():
    """include/edlib.h (the edlibAlign drop-in of construct.cc:190-199) is served by the same library."""
    text = open(os.path.join(ROOT, "include", "edlib.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(edlib[A-Za-z]+)\s*\(", text)))
    assert declared == ["edlibAlign", "edlibAlignmentToCigar", "edlibDefaultAlignConfig", "edlibFreeAlignResult",
                        "edlibNewAlignConfig"]
    L = hip.lib()
    for name in declared:
        assert hasattr(L, name), "missing export: " + name


def test_no_oracle_in_product():
    """The product package must not import / link / reference the oracle."""
    pkg = os.path.join(ROOT, "raven_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".sh")):
                text = open(os.path.join(dirpath, f)).read()
                assert "raven_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_fails_loudly_without_gpu():
    if hip.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(hip.RavenHipError) as ei:
        hip.Engine()
    assert "no HIP device" in str(ei.value)


def test_cpp_facade_programs_compile_and_link(tmp_path):
    """Every program under tests/cpp (the reference's call sequences against the header-only facades) compiles with g++
    and links against libraven_hip.so — no GPU needed; running them is tests/test_gpu_facade.py's job."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "raven_amd", "lib")
    for src in sorted(glob.glob(os.path.join(root, "tests", "cpp", "*.cpp"))):
        exe = str(tmp_path / os.path.basename(src)[:-4])
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), "-I",
                               os.path.join(root, "tests", "cpp"), "-o", exe, src, "-L", lib, "-lraven_hip",
                               "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
        assert os.path.exists(exe)
