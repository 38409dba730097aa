"""Host logic of the sharded pass (raven_amd/sharded.py): partitioning, regrouping, the global Filter cutoff, and
the communicator over gloo with world_size 2 (CPU)."""
import os
import subprocess
import sys

import numpy as np

from raven_amd import sharded
from tests import sharded_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_contiguous_and_balanced():
    rng = np.random.default_rng(0)
    lengths = rng.integers(1000, 60000, size=997).astype(np.uint32)
    for world in (1, 2, 3, 8):
        b = sharded.partition_reads(lengths, world)
        assert b[0] == 0 and b[-1] == lengths.shape[0] and np.all(np.diff(b) >= 0) and b.shape[0] == world + 1
        per = [int(lengths[b[i]:b[i + 1]].sum()) for i in range(world)]
        assert max(per) - min(per) <= 2 * int(lengths.max())
    assert sharded.partition_reads(np.zeros(0, np.uint32), 4).tolist() == [0, 0, 0, 0, 0]


def test_slice_reads_keeps_global_ids_and_bases():
    from raven_amd import synth
    g = synth.make_genome(30_000, seed=1)
    rs, _ = synth.make_reads(g, 5, 2000, seed=2)
    s = sharded.slice_reads(rs, 10, 25)
    assert s.ids.tolist() == list(range(10, 25)) and s.n == 15
    for i in range(15):
        assert s.inflate(i) == rs.inflate(10 + i)


def test_hash_owner_is_deterministic_and_spreads_small_values():
    v = np.arange(100_000, dtype=np.uint64)  # window minima are small values: must still spread
    for world in (2, 3, 8):
        o = sharded.hash_owner(v, world)
        assert np.array_equal(o, sharded.hash_owner(v.copy(), world)) and o.min() == 0 and o.max() == world - 1
        cnt = np.bincount(o, minlength=world)
        assert cnt.max() < 1.1 * cnt.mean()
    assert np.all(sharded.hash_owner(v, 1) == 0)


def test_regroup_by_read_merges_sources_per_read():
    rng = np.random.default_rng(3)
    n, srcs = 50, 3
    counts = [rng.integers(0, 5, size=n) for _ in range(srcs)]
    datas = []
    for s in range(srcs):
        read_of = np.repeat(np.arange(n), counts[s])
        a = (read_of * 1000 + s).astype(np.uint64)
        datas.append((a, a + np.uint64(7)))
    seg, (x, y) = sharded.regroup_by_read(counts, datas)
    assert int(seg[-1]) == sum(int(c.sum()) for c in counts) and np.array_equal(y, x + np.uint64(7))
    for i in range(n):
        got = x[int(seg[i]):int(seg[i + 1])]
        assert np.all(got // 1000 == i)
        assert sorted((got % 1000).tolist()) == sorted(sum(([s] * int(counts[s][i]) for s in range(srcs)), []))


def test_global_occurrence_equals_single_quantile():
    rng = np.random.default_rng(4)
    counts = np.concatenate([rng.geometric(0.3, size=20_000), rng.integers(100, 200_000, size=30)]).astype(np.uint32)
    for freq in (0.0, 0.001, 0.01, 0.5, 1.0):
        want = 0xFFFFFFFF if freq == 0 else int(np.sort(counts)[min(int((1 - freq) * counts.shape[0]), counts.shape[0] - 1)]) + 1
        for world in (1, 2, 3):
            perm = rng.permutation(counts.shape[0])
            parts = np.array_split(counts[perm], world)
            got = sharded_util.run_ranks(world, lambda r, comm: sharded.global_occurrence(parts[r], freq, comm))
            assert got == [want] * world, (freq, world, got, want)


WORKER = r"""
import numpy as np, torch.distributed as dist
from raven_amd import sharded
dist.init_process_group("gloo")
c = sharded.Comm(dist)
r, w = c.rank, c.world
dt = np.dtype([("a", "<u4"), ("b", "<u4"), ("c", "<u8")])
parts = []
for h in range(w):
    p = np.zeros(3 + r + 2 * h, dtype=dt)
    p["a"], p["b"], p["c"] = r, h, np.arange(p.shape[0]) + 2 ** 40
    parts.append(p)
got = c.all_to_all_v(parts)
for s in range(w):
    assert got[s].shape[0] == 3 + s + 2 * r and np.all(got[s]["a"] == s) and np.all(got[s]["b"] == r)
    assert np.array_equal(got[s]["c"], np.arange(got[s].shape[0]) + 2 ** 40)
assert c.all_reduce_sum(np.arange(5) * (r + 1)).tolist() == (np.arange(5) * sum(range(1, w + 1))).tolist()
assert c.all_gather_v(np.arange(r + 1)).tolist() == sum((list(range(s + 1)) for s in range(w)), [])
assert c.bytes_sent > 0
dist.destroy_process_group()
import sys
sys.stdout.write("COMM_OK_%d\n" % r)
sys.stdout.flush()
"""


def test_comm_over_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(sharded_util.free_port()), str(script)], capture_output=True,
                       text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "COMM_OK_0" in r.stdout and "COMM_OK_1" in r.stdout, r.stdout[-2000:]


WORKER_DEV = r"""
import torch, torch.distributed as dist
from raven_amd import sharded
dist.init_process_group("gloo")
c = sharded.DeviceComm(dist, device="cpu")   # the tensor collectives of the device-resident pass, on CPU tensors over gloo
r, w = c.rank, c.world
send = [2 + r + 3 * h for h in range(w)]
flat = torch.cat([torch.full((send[h],), 1000 * r + h, dtype=torch.int64) for h in range(w)] + [torch.zeros(5, dtype=torch.int64)])
out, lens = c.all_to_all_flat_t(flat, send)   # trailing slack after the parts is ignored
assert lens == [2 + s + 3 * r for s in range(w)], lens
o = 0
for s in range(w):
    assert torch.all(out[o:o + lens[s]] == 1000 * s + r)
    o += lens[s]
assert o == out.shape[0]
parts = c.all_to_all_t([torch.arange(1 + r + h, dtype=torch.int64) + 10 * r for h in range(w)])
for s in range(w):
    assert parts[s].tolist() == [x + 10 * s for x in range(1 + s + r)]
empty, elens = c.all_to_all_flat_t(torch.zeros(0, dtype=torch.int64), [0] * w)
assert empty.shape[0] == 0 and elens == [0] * w
assert c.bytes_sent > 0
dist.destroy_process_group()
import sys
sys.stdout.write("DEVCOMM_OK_%d\n" % r)
sys.stdout.flush()
"""


def test_device_comm_tensor_exchanges_over_gloo_world2(tmp_path):
    script = tmp_path / "worker_dev.py"
    script.write_text(WORKER_DEV)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(sharded_util.free_port()), str(script)], capture_output=True,
                       text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DEVCOMM_OK_0" in r.stdout and "DEVCOMM_OK_1" in r.stdout, r.stdout[-2000:]


def test_index_batches_follow_the_reference_schedule():
    """construct.cc:32-37: a batch closes with the read that brings its bases to the limit, or with the last read."""
    lengths = np.array([10, 10, 10, 10, 10, 10, 10], dtype=np.uint32)
    assert sharded.index_batches(lengths, 25) == [(0, 3), (3, 6), (6, 7)]
    assert sharded.index_batches(lengths, 1 << 32) == [(0, 7)]
    assert sharded.index_batches(lengths, 1) == [(i, i + 1) for i in range(7)]


def test_flush_windows_follow_the_reference_schedule():
    # construct.cc:56-70: bytes += len; flush when bytes >= limit or at the last read
    L = np.array([5, 5, 5, 5, 5], dtype=np.uint32)
    assert sharded.flush_windows(L, 10) == [(0, 2), (2, 4), (4, 5)]
    assert sharded.flush_windows(L, 11) == [(0, 3), (3, 5)]
    assert sharded.flush_windows(L, 1000) == [(0, 5)]
    assert sharded.flush_windows(L, 1) == [(i, i + 1) for i in range(5)]
    assert sharded.flush_windows(np.zeros(0, np.uint32), 10) == []


def test_polish_round_sharded_host_logic_with_a_stub_engine():
    """The host side of the sharded polishing round (read slices for the mapping, all-gather of the best-overlap table,
    window ranges, gather of the consensus pieces) with a stub in place of the engine: every rank must hand the SAME
    complete table to the round, ask for ITS window range, and the pieces must come back in rank and target order."""
    from raven_amd import seqio

    class Handle:
        def __init__(self, n):
            self.n = n

    n_reads, world, w = 37, 3, 500
    lengths = np.array([1200, 499, 2001], dtype=np.uint32)  # 3 + 1 + 5 = 9 windows
    targets_rs = seqio.ReadSet(packed=np.zeros(1, np.uint64), word_offsets=np.zeros(4, np.uint64), lengths=lengths,
                               ids=np.arange(3, dtype=np.uint32))
    rng = np.random.default_rng(5)
    full_best = rng.integers(0, 1 << 31, size=(n_reads, 8), dtype=np.uint32)
    full_bt = rng.integers(0, 3, size=n_reads).astype(np.uint32)
    full_bt[::7] = 0xFFFFFFFF
    seen = {}

    class StubEngine:
        def __init__(self, rank):
            self.rank, self.table = rank, None

        def polish_map_best(self, targets, reads, first, last, err=0.3):
            seen.setdefault("slices", {})[self.rank] = (first, last)
            return full_best[first:last], full_bt[first:last], 0

        def polish_set_best(self, best, bt):
            self.table = (np.array(best, copy=True), np.array(bt, copy=True))

        def polish_round_range(self, targets, reads, lo, hi, **kw):
            assert self.table is not None and np.array_equal(self.table[0], full_best) and np.array_equal(self.table[1], full_bt)
            seen.setdefault("ranges", {})[self.rank] = (lo, hi)
            # the piece of target t = one byte per window of t inside [lo, hi), valued by the global window index
            first = np.concatenate([[0], np.cumsum((lengths.astype(np.int64) + w - 1) // w)])
            cons, nw, npol = [], np.zeros(3, np.uint32), np.zeros(3, np.uint32)
            for t in range(3):
                a, b = max(lo, int(first[t])), min(hi, int(first[t + 1]))
                cons.append(np.arange(a, max(a, b), dtype=np.uint8))
                nw[t] = max(0, b - a)
                npol[t] = max(0, b - a - (1 if t == 1 else 0))
            return cons, nw, npol, {}

    def rank_fn(r, comm):
        return sharded.polish_round_sharded(StubEngine(r), Handle(3), Handle(n_reads), comm, targets_rs, w=w)

    res = sharded_util.run_ranks(world, rank_fn)
    assert sorted(seen["slices"].values()) == [(0, 12), (12, 24), (24, 37)]
    assert sorted(seen["ranges"].values()) == [(0, 3), (3, 6), (6, 9)]
    for cons, ratio in res:
        assert [c.tolist() for c in cons] == [[0, 1, 2], [3], [4, 5, 6, 7, 8]]
        assert np.allclose(ratio, [1.0, 0.0, 1.0])
