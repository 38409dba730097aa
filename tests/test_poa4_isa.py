"""The generated code of the NW step of the window-consensus kernel (raven_amd/csrc/poa4.hip), checked without a GPU.

Round 5 found `s_waitcnt vmcnt(0)` inside the row-switch block of every NW step: a register reloaded from scratch right
before the loop stayed "in flight" for the compiler's wait-count bookkeeping, so its first use in the loop waited for every
outstanding memory operation — the descriptor prefetch of the same service point and the backpointer store — in nearly
every step (+15 % on the NW).  Nothing in the source shows it and any change of register pressure
can bring it back; the kernel marks its steps in the assembly (P4_MARK), so the check is a count."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_vector_memory_wait_inside_an_nw_step(tmp_path):
    out = tmp_path / "poa4.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "--cuda-device-only", "-S",
                        os.path.join(ROOT, "raven_amd", "csrc", "poa4.hip"), "-I", os.path.join(ROOT, "include"), "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    steps, cur = [], None
    for line in out.read_text().split("\n"):
        if "P4MARK step_begin" in line:
            cur = []
        elif "P4MARK step_end" in line and cur is not None:
            steps.append(cur)
            cur = None
        elif cur is not None:
            t = line.strip()
            if t and not t.startswith((";", ".")):
                cur.append(t)
    assert len(steps) == 8, len(steps)  # the loop is unrolled over the 8 steps between two service points
    for ins in steps:
        waits = [t for t in ins if re.match(r"s_waitcnt\s+vmcnt", t)]
        assert not waits, waits
        assert not [t for t in ins if "scratch_" in t]            # no spill traffic in a step either
        assert len([t for t in ins if t.startswith("ds_")]) <= 10  # 4 + 4 in-edge reads, the end-node read, the own row's store
