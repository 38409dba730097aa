"""N engines behind one host process (rvn_group_*, raven_amd/csrc/group.hip): the C++ driver of the sharded pass and the
sharded polishing round — SURVEY 8(b)'s engine with a device list, 8(e)'s exchanges, without Python or torch in the loop.
tests/cpp/group_test.cpp runs 2 and 3 virtual ranks on the one GPU of the test box and compares every rank's slice with
the single-engine result itself; here its verdicts are checked."""
import subprocess

import numpy as np
import pytest

from raven_amd import synth
from tests.test_gpu_facade import _build, _write_reads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_ranks,flush,index_batch", [(2, 1 << 30, 1 << 32), (3, 400_000, 1 << 32), (3, 400_000, 1_700_000)])
def test_group_pass_and_round_are_bit_identical_to_the_single_engine(tmp_path, n_ranks, flush, index_batch):
    exe = _build(tmp_path, "group_test")
    g = synth.make_genome(250_000, seed=41)
    rs, _ = synth.make_reads(g, 20, 6000, seed=42)
    rpath = _write_reads(tmp_path, rs)
    drafts = [synth.make_draft(g[:120_000], seed=43), synth.make_draft(g[120_000:], seed=44)]
    dpath = str(tmp_path / "drafts.txt")
    with open(dpath, "wb") as f:
        for d in drafts:
            f.write(bytes(np.frombuffer(b"ACGT", np.uint8)[d]) + b"\n")
    # index_batch 1.7 Mb on 5 Mb of reads: three index batches (construct.cc:32-37), every read up to a batch's end mapped
    # against it — cut inside a rank's read range
    r = subprocess.run([exe, rpath, dpath, str(n_ranks), str(flush), str(index_batch)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().split("\n")
    assert int(lines[0].split()[-1]) > 5000                       # the single pass found overlaps
    bounds = [int(x) for x in lines[1].split()[1:]]
    assert bounds[0] == 0 and bounds[-1] == rs.n and len(bounds) == n_ranks + 1 and all(np.diff(bounds) > 0)
    ranks = [ln for ln in lines if ln.startswith("rank ")]
    assert len(ranks) == n_ranks and all(ln.endswith("identical 1") for ln in ranks), ranks
    targets = [ln for ln in lines if ln.startswith("target ")]
    assert len(targets) == 2 and all(ln.endswith("identical 1") for ln in targets), targets
    assert all(float(ln.split()[5]) > 0.99 for ln in targets)     # (nearly) every window polished
    # the same round with block qualities (the FASTQ variant through the group): byte-identical to the
    # single engine (a fifth of the reads sit below q = 10: their layers fail the mean-quality filter on every rank alike)
    qt = [ln for ln in lines if ln.startswith("quality target ")]
    assert len(qt) == 2 and all(" identical 1 " in ln for ln in qt), qt
    assert all(float(ln.split()[6]) > 0.9 for ln in qt), qt
    # virtual ranks share one device: every pair reaches the other's memory directly
    assert "peer access all 1" in lines


def test_device_group_facade_equals_the_single_device_templates(tmp_path):
    """raven::FindOverlapsAndCreatePiles<Pile> / raven::PolishRound over a raven::DeviceGroup
    (include/raven_hip/multi_gpu.hpp) against the single-device facades on the same input."""
    exe = _build(tmp_path, "multi_gpu_facade_test")
    g = synth.make_genome(150_000, seed=61)
    rs, _ = synth.make_reads(g, 18, 5000, seed=62)
    rpath = _write_reads(tmp_path, rs)
    dpath = str(tmp_path / "drafts.txt")
    with open(dpath, "wb") as f:
        for d in (synth.make_draft(g[:70_000], seed=63), synth.make_draft(g[70_000:], seed=64)):
            f.write(bytes(np.frombuffer(b"ACGT", np.uint8)[d]) + b"\n")
    r = subprocess.run([exe, rpath, dpath, "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().split("\n")
    a = lines[0].split()
    assert a[:2] == ["ranks", "2"] and int(a[3]) > 2000 and a[5] == "0", lines[0]
    assert lines[1] == "polished 2 differing_targets 0"
