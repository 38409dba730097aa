"""Sharded single-genome FindOverlapsAndCreatePiles (raven_amd/sharded.py over the rvn_shard_* stages) against
the single-GPU pass: every rank's slice of pile coverage and truncated overlap lists must be BIT-IDENTICAL.
Virtual ranks = threads with one engine each on the one GPU of the test box; plus a real two-process run over
torch.distributed (gloo) sharing that GPU — on a multi-GPU node the same code runs one rank per GPU over RCCL."""
import os
import subprocess
import sys

import numpy as np
import pytest

from raven_amd import hip, sharded, synth
from tests import sharded_util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single(rs, **kw):
    eng = hip.Engine(15, 5)
    p = eng.find_overlaps_and_create_piles(eng.upload(rs), **kw)
    data, poff = p.piles()
    kept, koff = p.overlaps()
    occ = eng.occurrence
    p.close()
    return data, poff, kept, koff, occ


@pytest.mark.parametrize("world", [1, 2, 3, 4])
@pytest.mark.parametrize("use_minhash", [False, True])
def test_sharded_pass_is_bit_identical(world, use_minhash):
    g = synth.make_genome(300_000, seed=61)
    rs, _ = synth.make_reads(g, 20, 6000, seed=62)
    data, poff, kept, koff, occ = _single(rs, use_minhash=use_minhash)
    assert kept.shape[0] > 5000

    def rank_fn(r, comm):
        return sharded.find_overlaps_and_create_piles_sharded(hip.Engine(15, 5), rs, comm, use_minhash=use_minhash)

    res = sharded_util.run_ranks(world, rank_fn)
    assert [x["lo"] for x in res] + [res[-1]["hi"]] == sharded.partition_reads(rs.lengths, world).tolist()
    for x in res:
        assert x["occurrence"] == occ
        sharded_util.check_against_single(x, data, poff, kept, koff)
    if world > 1:
        assert sum(x["stats"]["matches_sent"] for x in res) > 0 and sum(x["stats"]["overlaps_sent"] for x in res) > 0


def test_sharded_pass_repeats_and_small_kmax():
    """Repeats exercise the global Filter cutoff (hash classes see different key-count distributions)."""
    rng = np.random.default_rng(5)
    g = synth.make_genome(200_000, seed=71)
    rep = g[1000:6000].copy()
    for at in rng.integers(10_000, 190_000, size=12):
        g[at:at + 5000] = rep
    rs, _ = synth.make_reads(g, 15, 5000, seed=72)
    data, poff, kept, koff, occ = _single(rs, freq=0.01, kmax=4)
    res = sharded_util.run_ranks(3, lambda r, comm: sharded.find_overlaps_and_create_piles_sharded(
        hip.Engine(15, 5), rs, comm, freq=0.01, kmax=4))
    assert occ < 0xFFFFFFFF
    for x in res:
        assert x["occurrence"] == occ
        sharded_util.check_against_single(x, data, poff, kept, koff)


WORKER = r"""
import numpy as np, torch.distributed as dist
from raven_amd import hip, sharded, synth
from tests import sharded_util
dist.init_process_group("gloo")
g = synth.make_genome(250_000, seed=81)
rs, _ = synth.make_reads(g, 20, 6000, seed=82)
eng = hip.Engine(15, 5, device=0)   # both ranks share the one GPU of the test box
res = sharded.find_overlaps_and_create_piles_sharded(eng, rs, sharded.Comm(dist))
ref = hip.Engine(15, 5, device=0)
p = ref.find_overlaps_and_create_piles(ref.upload(rs))
data, poff = p.piles(); kept, koff = p.overlaps()
sharded_util.check_against_single(res, data, poff, kept, koff)
assert res["occurrence"] == ref.occurrence and res["stats"]["bytes_sent"] > 0
rank = dist.get_rank()
dist.barrier(); dist.destroy_process_group()
import sys
sys.stdout.write("SHARD_OK_%d lo=%d hi=%d kept=%d\n" % (rank, res["lo"], res["hi"], res["overlaps"].shape[0]))
sys.stdout.flush()
"""


def test_sharded_pass_two_processes_over_torch_distributed(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(sharded_util.free_port()), str(script)], capture_output=True,
                       text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "SHARD_OK_0" in r.stdout and "SHARD_OK_1" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("world", [1, 3])
def test_device_resident_sharded_pass_is_bit_identical(world):
    """Same pass with every exchange buffer in HBM (torch CUDA tensors, rvn_shard_*_dev)."""
    import torch
    dev = torch.device("cuda", 0)
    g = synth.make_genome(300_000, seed=91)
    rs, _ = synth.make_reads(g, 20, 6000, seed=92)
    data, poff, kept, koff, occ = _single(rs)

    def rank_fn(r, comm):
        return sharded.find_overlaps_and_create_piles_sharded_dev(hip.Engine(15, 5), rs, comm, dev)

    res = sharded_util.run_ranks(world, rank_fn)
    for x in res:
        assert x["occurrence"] == occ
        sharded_util.check_against_single(x, data, poff, kept, koff)
    if world > 1:
        assert sum(x["stats"]["matches_sent"] for x in res) > 0 and sum(x["stats"]["overlaps_sent"] for x in res) > 0


def test_hash_owner_torch_equals_numpy():
    import torch
    v = np.random.default_rng(1).integers(0, 1 << 62, size=10_000, dtype=np.uint64)
    v[:100] = np.arange(100, dtype=np.uint64)
    for world in (2, 3, 8):
        a = sharded.hash_owner(v, world)
        b = sharded.hash_owner_t(torch.from_numpy(v.view(np.int64)).cuda(), world).cpu().numpy()
        assert np.array_equal(a, b)


@pytest.mark.parametrize("world", [2, 3])
def test_polishing_round_sharded_by_windows_is_byte_identical(world):
    from raven_amd import seqio
    from tests import polish_util
    truths, drafts, targets_rs, reads_rs, _ = polish_util.make_case(genome_len=40_000, coverage=20, read_len=3000, seed=31,
                                                                    n_targets=3)
    eng0 = hip.Engine(15, 5)
    ref, ref_ratio, _ = eng0.polish_round(eng0.upload(targets_rs), eng0.upload(reads_rs))

    def rank_fn(r, comm):
        eng = hip.Engine(15, 5)
        return sharded.polish_round_sharded(eng, eng.upload(targets_rs), eng.upload(reads_rs), comm, targets_rs)

    res = sharded_util.run_ranks(world, rank_fn)
    for cons, ratio in res:
        assert np.allclose(ratio, ref_ratio)
        for t in range(3):
            assert np.array_equal(cons[t], ref[t])


@pytest.mark.parametrize("variant", ["host", "device"])
def test_sharded_pass_with_several_flush_windows(variant):
    """A pass whose query reads are flushed in several windows (2^30 bases each in the reference; 300 kb here): piles
    persist across the windows, truncation happens per flush — the result depends on the flush schedule and must match
    the single-GPU pass run with the same one."""
    import torch
    g = synth.make_genome(200_000, seed=101)
    rs, _ = synth.make_reads(g, 25, 5000, seed=102)
    flush = 300_000
    assert len(sharded.flush_windows(rs.lengths, flush)) >= 10
    data, poff, kept, koff, occ = _single(rs, kmax=8, flush_bases=flush)
    one_flush = _single(rs, kmax=8)
    assert not np.array_equal(one_flush[2], kept)          # the schedule does change the result ...

    def rank_fn(r, comm):
        eng = hip.Engine(15, 5)
        if variant == "host":
            return sharded.find_overlaps_and_create_piles_sharded(eng, rs, comm, kmax=8, flush_bases=flush)
        return sharded.find_overlaps_and_create_piles_sharded_dev(eng, rs, comm, torch.device("cuda", 0), kmax=8,
                                                                  flush_bases=flush)

    for world in (1, 3):
        for x in sharded_util.run_ranks(world, rank_fn):
            assert x["occurrence"] == occ
            sharded_util.check_against_single(x, data, poff, kept, koff)   # ... and the sharded pass follows it
