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


WORKER_RCCL = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
from raven_amd import hip, sharded, synth, seqio
from tests import sharded_util
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))    # RCCL on ROCm
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", 0)
g = synth.make_genome(5_000_000, seed=81)                              # BASELINE configs[2] size
rs, _ = synth.make_reads(g, 30, 10000, seed=82)
eng = hip.Engine(15, 5, device=0)
comm = sharded.DeviceComm(dist, device="cuda", force=True)
assert comm.dist is not None
res = sharded.find_overlaps_and_create_piles_sharded_dev(eng, rs, comm, dev)   # all_to_all_single / all_reduce over RCCL
ref = hip.Engine(15, 5, device=0)
p = ref.find_overlaps_and_create_piles(ref.upload(rs))
data, poff = p.piles(); kept, koff = p.overlaps()
sharded_util.check_against_single(res, data, poff, kept, koff)
assert res["occurrence"] == ref.occurrence
# collectives as the pass uses them, on CUDA tensors, through the nccl backend
t = torch.arange(1000, dtype=torch.int64, device=dev)
out = comm.all_to_all_t([t])
assert torch.equal(out[0], t)
assert int(comm.all_reduce_sum(np.array([7, 9]))[1]) == 9
assert np.array_equal(comm.all_gather_v(np.arange(5)), np.arange(5))
# one polishing round through the sharded entry points (reads by slice, windows by range, all-gather of the pieces)
draft = synth.make_draft(g[:200_000], seed=83)
peng = hip.Engine(15, 5, device=0)
preads = peng.upload(rs)
targets = peng.upload_codes([draft])
cons_s, ratio_s = sharded.polish_round_sharded(peng, targets, preads, comm, targets.rs)
cons_1, ratio_1, _ = peng.polish_round(targets, preads)
assert len(cons_s) == len(cons_1) and all(np.array_equal(a, b) for a, b in zip(cons_s, cons_1))
dist.barrier(); dist.destroy_process_group()
sys.stdout.write("RCCL_OK kept=%d\n" % res["overlaps"].shape[0]); sys.stdout.flush()
"""


def test_sharded_pass_over_rccl_world_size_one(tmp_path):
    """The nccl (= RCCL) backend path itself, on the one GPU a test box has: world size 1 under torch.distributed.run,
    BASELINE configs[2] size, every exchange of the sharded pass and of the sharded polishing round through
    all_to_all_single / all_reduce on CUDA tensors; results bit-identical to the fused single-GPU calls.  Then bench.py
    in the same launch mode prints its JSON line with the exchange volume."""
    script = tmp_path / "worker_rccl.py"
    script.write_text(WORKER_RCCL)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(sharded_util.free_port()), str(script)], capture_output=True,
                       text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "RCCL_OK" in r.stdout, r.stdout[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(sharded_util.free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sharded", "--workload", "c2", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--load-bases", "0"], capture_output=True, text=True,
                       timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    import json
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["collectives"] == "nccl"
    assert line["exchange_bytes_per_step"] is not None


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


@pytest.mark.parametrize("variant", ["host", "device"])
@pytest.mark.parametrize("use_minhash", [False, True])
def test_sharded_pass_with_several_index_batches(variant, use_minhash):
    """More than one index batch (2^32 bases each in the reference, construct.cc:32-37; 1.6 Mb here on 5 Mb of reads, so three
    batches whose ends fall inside ranks' read ranges): per batch the members' minimizers and, as query-only entries, the
    minhash-selected minimizers of every earlier read go to the hash owners; every read up to the batch's end is mapped
    against the shard in flush windows.  Filter runs per batch on the members alone.  Bit-identical to the single-GPU pass
    with the same batch size, and different from the one-batch result."""
    import torch
    g = synth.make_genome(250_000, seed=131)
    rs, _ = synth.make_reads(g, 20, 5000, seed=132)
    batch, flush = 1_600_000, 700_000
    assert len(sharded.index_batches(rs.lengths, batch)) >= 3
    data, poff, kept, koff, occ = _single(rs, kmax=16, flush_bases=flush, index_batch_bases=batch, use_minhash=use_minhash)
    one_batch = _single(rs, kmax=16, flush_bases=flush, use_minhash=use_minhash)
    assert not np.array_equal(one_batch[2], kept)          # the batch schedule does change the result ...

    def rank_fn(r, comm):
        eng = hip.Engine(15, 5)
        if variant == "host":
            return sharded.find_overlaps_and_create_piles_sharded(eng, rs, comm, kmax=16, flush_bases=flush,
                                                                  index_batch_bases=batch, use_minhash=use_minhash)
        return sharded.find_overlaps_and_create_piles_sharded_dev(eng, rs, comm, torch.device("cuda", 0), kmax=16,
                                                                  flush_bases=flush, index_batch_bases=batch,
                                                                  use_minhash=use_minhash)

    for world in (1, 2, 3):
        for x in sharded_util.run_ranks(world, rank_fn):
            assert x["occurrence"] == occ                  # (the last batch's cutoff, as the single engine leaves it)
            sharded_util.check_against_single(x, data, poff, kept, koff)   # ... and the sharded pass follows it


@pytest.mark.parametrize("world", [1, 3, 8])
def test_partition_and_regroup_kernels_equal_the_host_formulas(world):
    """shard.hip against the numpy statements of the same steps (raven_amd/sharded.py host variant)."""
    import torch
    dev = torch.device("cuda", 0)
    eng = hip.Engine(15, 5)
    rng = np.random.default_rng(100 + world)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    # minimizers by hash class, stable
    n = 100_003
    val = rng.integers(0, 1 << 40, size=n, dtype=np.uint64)
    val[:500] = np.arange(500, dtype=np.uint64)
    org = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) | (rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63))
    v_d, o_d = t(val), t(org)
    v_o, o_o = torch.empty_like(v_d), torch.empty_like(o_d)
    torch.cuda.synchronize()
    cnt = eng.shard_split_minimizers_dev(v_d.data_ptr(), o_d.data_ptr(), n, world, v_o.data_ptr(), o_o.data_ptr())
    owner = sharded.hash_owner(val, world)
    order = np.argsort(owner, kind="stable")
    assert cnt == np.bincount(owner, minlength=world).tolist()
    assert np.array_equal(v_o.cpu().numpy().view(np.uint64), val[order])
    assert np.array_equal(o_o.cpu().numpy().view(np.uint64), org[order])
    assert eng.shard_count_flagged_dev(o_d.data_ptr(), n) == int((org >> np.uint64(63)).sum())
    # overlaps by the owner of the rhs read, own ones stay
    n_reads, m = 5000, 40_001
    bounds = np.unique(np.concatenate([[0, n_reads], rng.integers(1, n_reads, size=world - 1)])).astype(np.uint32)
    while bounds.shape[0] < world + 1:  # duplicates collapsed: pad with empty trailing ranges
        bounds = np.concatenate([bounds, [n_reads]]).astype(np.uint32)
    ovl = rng.integers(0, 1 << 20, size=(m, 8), dtype=np.uint32)
    ovl[:, 3] = rng.integers(0, n_reads, size=m)
    me = world // 2
    o_dv = torch.from_numpy(ovl.view(np.int64)).to(dev)
    o_out = torch.empty_like(o_dv)
    torch.cuda.synchronize()
    oc = eng.shard_split_overlaps_dev(o_dv.data_ptr(), m, bounds, world, me, o_out.data_ptr())
    rhs_owner = np.searchsorted(bounds, ovl[:, 3], side="right") - 1
    rhs_owner = np.minimum(rhs_owner, world - 1)
    want = np.concatenate([ovl[rhs_owner == h] for h in range(world) if h != me] + [ovl[rhs_owner == me]])
    key = np.where(rhs_owner == me, world, rhs_owner)
    assert oc == np.bincount(key, minlength=world + 1).tolist()
    got = o_out.cpu().numpy().view(np.uint32).reshape(m, 8)
    want2 = ovl[np.argsort(key, kind="stable")]
    assert np.array_equal(got, want2) and want.shape == want2.shape
    # regroup of the matches of every source per read
    nr = 3001
    cnts = [rng.integers(0, 6, size=nr).astype(np.int64) for _ in range(world)]
    datas = [(rng.integers(0, 1 << 62, size=int(c.sum()), dtype=np.int64), rng.integers(0, 1 << 62, size=int(c.sum()), dtype=np.int64))
             for c in cnts]
    seg_w, (g_w, p_w) = sharded.regroup_by_read(cnts, datas)
    c_d = [t(c) for c in cnts]
    g_d = [t(d[0]) for d in datas]
    p_d = [t(d[1]) for d in datas]
    total = int(seg_w[-1])
    seg_o = torch.empty(nr + 1, dtype=torch.int64, device=dev)
    g_o, p_o = torch.empty(total, dtype=torch.int64, device=dev), torch.empty(total, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    eng.shard_regroup_dev([x.data_ptr() for x in c_d], [x.data_ptr() for x in g_d], [x.data_ptr() for x in p_d],
                          [int(c.sum()) for c in cnts], nr, seg_o.data_ptr(), g_o.data_ptr(), p_o.data_ptr())
    assert np.array_equal(seg_o.cpu().numpy().view(np.uint64), seg_w)
    assert np.array_equal(g_o.cpu().numpy(), g_w) and np.array_equal(p_o.cpu().numpy(), p_w)
    # adjacent differences
    cnt_o = torch.empty(nr, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    eng.shard_adjacent_diff_dev(seg_o.data_ptr(), nr, cnt_o.data_ptr())
    assert np.array_equal(cnt_o.cpu().numpy(), np.diff(seg_w.astype(np.int64)))


_FOREIGN_PIECES = r"""
import os, sys, json
import numpy as np
from raven_amd import hip, synth
g = synth.make_genome(120_000, seed=31)
rs, _ = synth.make_reads(g, 12, 4000, seed=32)
eng = hip.Engine(15, 5)
own = eng.upload(rs)
n = own.n
def foreign():
    k = eng.shard_sketch_range_count(own, 0, n, True, foreign=True)
    v, o = eng.shard_sketch_fetch(k)
    return np.array(v), np.array(o)
v1, o1 = foreign()
os.environ["RVN_FOREIGN_PIECE_BASES"] = "150000"      # ~10 pieces of this read set
v2, o2 = foreign()
del os.environ["RVN_FOREIGN_PIECE_BASES"]
same = bool(np.array_equal(v1, v2) and np.array_equal(o1, o2))
flags = bool(len(o1) and np.all((o1 >> np.uint64(62)) == 3))
os.environ["RVN_SKETCH_LIMIT"] = "1000"               # what 2^32 is to a 13-Gbase range
err = ""
try:
    eng.shard_sketch_range_count(own, 0, n, False, foreign=False)
except ValueError as ex:
    err = str(ex)
del os.environ["RVN_SKETCH_LIMIT"]
k = eng.shard_sketch_range_count(own, 0, n, False, foreign=False)   # the engine is usable after the refusal
print(json.dumps({"same": same, "flags": flags, "n": int(len(o1)), "err": err, "after": int(k)}))
"""


@pytest.mark.gpu
def test_query_only_sketch_in_pieces_and_the_refusal_of_a_sketch_beyond_32_bit_offsets(tmp_path):
    """the reads of earlier index batches are sketched as ONE range per batch, whose raw minimizers have no bound
    (>= 2^32 from ~12.9 Gbases at w = 5: the sketch's 32-bit offsets would wrap silently).  rvn_shard_sketch_range takes the
    query-only range in base-bounded pieces and appends the selected entries — identical to the one-piece result — and a
    sketch whose total would not fit is refused with RVN_EINVAL, not computed modulo 2^32.  Debug library: the piece size and
    the limit are lowered by environment switches that exist only there."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "raven_amd", "lib", "libraven_hip_test.so")
    env = dict(os.environ, RVN_LIB_PATH=lib, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", _FOREIGN_PIECES], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
    assert d["same"] and d["flags"] and d["n"] > 1000, d
    assert "2^32" in d["err"] and "pieces" in d["err"], d
    assert d["after"] > 1000, d
