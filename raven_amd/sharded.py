"""Sharded single-genome FindOverlapsAndCreatePiles (construct.cc:14-121 split across GPUs).

One process per GPU.  Reads are range-partitioned by pile (rank g owns a contiguous id range, balanced by bases, and
uploads only those reads); minimizer values are partitioned by hash class.  Three exchanges over
``torch.distributed`` (RCCL all-to-all on a multi-GPU node, gloo in the tests):

  1. every minimizer (value, origin+query flag) to the owner of its hash class  -> owner builds its index shard
     (all-reduce of the per-key count histogram -> the exact global ``Filter`` cutoff on every rank)
  2. every match (candidate pair) from the hash owner's self-join to the owner of the query read -> chaining
  3. every overlap also to the owner of its rhs read -> merge, ``AddLayers``, top-kMax truncation per pile

The compute of every stage is the same device code as the single-GPU pass (``rvn_shard_*`` in raven_hip.h); this file
is only the partitioning and the exchanges.  Result for the reads a rank owns: bit-identical to the single-GPU pass
(tests/test_gpu_sharded.py), because (a) pieces are concatenated in source-rank order = global (read, position)
order, which is the order ram's stable sort and the reference's serial merge see, and (b) the order of a read's
matches does not matter (total-order sorts follow).  Flush windows (2^30 bases of query reads each, construct.cc:56-70)
are processed one after the other — join of the window's query reads, exchange, chain, exchange, merge + AddLayers +
truncation into piles that persist across the windows — exactly as the reference flushes, so a pass with several
windows (configs[3]/[4]) is bit-identical as well; one index batch (total bases < 2^32) is the limit.  Two variants: `find_overlaps_and_create_piles_sharded` stages the
exchange buffers through host memory (numpy; gloo or RCCL), `find_overlaps_and_create_piles_sharded_dev` keeps them in
HBM as torch CUDA tensors (RCCL) — initialise torch.cuda before creating engines in that process.
"""
from __future__ import annotations

import numpy as np

from . import seqio

_MIX = np.uint64(0x9E3779B97F4A7C15)


def partition_reads(lengths: np.ndarray, world: int) -> np.ndarray:
    """Contiguous read ranges balanced by bases: bounds[world + 1]."""
    n = int(lengths.shape[0])
    cum = np.concatenate([[0], np.cumsum(lengths.astype(np.uint64))]).astype(np.float64)
    targets = cum[-1] * np.arange(1, world) / world
    inner = np.searchsorted(cum, targets, side="left")
    bounds = np.concatenate([[0], inner, [n]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def slice_reads(rs: seqio.ReadSet, lo: int, hi: int) -> seqio.ReadSet:
    """Reads [lo, hi) as their own packed set; ids stay the GLOBAL read indices."""
    w0, w1 = int(rs.word_offsets[lo]), int(rs.word_offsets[hi])
    return seqio.ReadSet(packed=np.ascontiguousarray(rs.packed[w0:w1]),
                         word_offsets=(rs.word_offsets[lo:hi + 1] - np.uint64(w0)).astype(np.uint64),
                         lengths=np.ascontiguousarray(rs.lengths[lo:hi]),
                         ids=np.arange(lo, hi, dtype=np.uint32))


def flush_windows(lengths: np.ndarray, flush_bases: int):
    """Query windows of a pass as the reference flushes them (construct.cc:56-70): a window closes with the read that
    brings its bases to flush_bases, or with the last read."""
    out, first, acc = [], 0, 0
    n = int(lengths.shape[0])
    for k in range(n):
        acc += int(lengths[k])
        if k != n - 1 and acc < flush_bases:
            continue
        out.append((first, k + 1))
        first, acc = k + 1, 0
    return out


def index_batches(lengths: np.ndarray, index_batch_bases: int):
    """Index batches of a pass as the reference cuts them (construct.cc:32-37): a batch closes with the read that brings
    its bases to index_batch_bases (2^32 there), or with the last read.  Every read up to a batch's end is mapped against
    it (:59-64), so the flush windows of batch [first, last) are flush_windows(lengths[:last])."""
    return flush_windows(lengths, index_batch_bases)


def hash_owner(values: np.ndarray, world: int) -> np.ndarray:
    """Owner rank of a minimizer value (multiplicative mix: window minima are skewed towards small values)."""
    if world == 1:
        return np.zeros(values.shape[0], dtype=np.int64)
    with np.errstate(over="ignore"):
        h = (values.astype(np.uint64) * _MIX) >> np.uint64(33)
    return (h % np.uint64(world)).astype(np.int64)


def global_occurrence(key_counts: np.ndarray, freq: float, comm) -> int:
    """ram Filter over ALL hash classes: (value at index (1-f)*U of the sorted per-key counts) + 1."""
    if freq == 0:
        return 0xFFFFFFFF
    hist = np.bincount(np.minimum(key_counts, 65535), minlength=65536).astype(np.int64)
    return occurrence_from_histogram(hist, key_counts[key_counts >= 65535].astype(np.int64), freq, comm)


def occurrence_from_histogram(hist: np.ndarray, over: np.ndarray, freq: float, comm) -> int:
    """Same from this rank's count-of-counts (bins 0..65534, bin 65535 = #keys with count >= 65535, listed in `over`):
    all-reduce of the histogram, all-gather of the overflow counts."""
    if freq == 0:
        return 0xFFFFFFFF
    over = comm.all_gather_v(np.asarray(over, dtype=np.int64))
    hist = comm.all_reduce_sum(np.asarray(hist, dtype=np.int64))
    u = int(hist.sum())
    if u == 0:
        return 0xFFFFFFFF
    nth = min(int((1 - freq) * u), u - 1)
    cum = np.cumsum(hist)
    c = int(np.searchsorted(cum, nth + 1, side="left"))
    if c >= 65535:
        below = int(cum[65534])
        c = int(np.sort(over)[nth - below])
    return c + 1


def regroup_by_read(counts_per_src, data_per_src):
    """Received per-source (per-read counts, flat arrays...) -> (per-read offsets, arrays grouped by read)."""
    n = counts_per_src[0].shape[0]
    total = np.zeros(n, dtype=np.int64)
    for c in counts_per_src:
        total += c.astype(np.int64)
    seg = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(total, out=seg[1:])
    outs = [np.zeros(int(seg[-1]), dtype=d.dtype) for d in data_per_src[0]]
    start = seg[:-1].astype(np.int64).copy()
    for c, datas in zip(counts_per_src, data_per_src):
        c = c.astype(np.int64)
        m = int(c.sum())
        if m:
            src_off = np.concatenate([[0], np.cumsum(c)[:-1]])
            dest = np.repeat(start - src_off, c) + np.arange(m, dtype=np.int64)
            for o, d in zip(outs, datas):
                o[dest] = d
        start += c
    return seg, outs


class Comm:
    """Variable-size exchanges of numpy arrays over torch.distributed (None / world 1 = identity)."""

    def __init__(self, dist=None, device="cpu", force=False):
        # force: go through the collectives at world size 1 as well (the backend's self-exchange: how a one-GPU box
        # exercises the RCCL path)
        self.dist = dist if (dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force)) else None
        self.device = device
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        self.bytes_sent = 0

    def _t(self, a):
        import torch
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def all_to_all_v(self, parts):
        """parts[h] = array for rank h (all the same dtype, itemsize multiple of 8 or int64/uint64).  Returns the list
        of arrays received, indexed by source rank."""
        if self.dist is None:
            return [parts[0]]
        import torch
        dt = parts[0].dtype
        words = [np.ascontiguousarray(p).view(np.int64).reshape(-1) for p in parts]
        per = dt.itemsize // 8
        cnt_in = torch.tensor([w.shape[0] for w in words], dtype=torch.int64, device=self.device)
        cnt_out = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        self.dist.all_to_all_single(cnt_out, cnt_in)
        cnt_out_l = [int(x) for x in cnt_out.tolist()]
        inp = self._t(np.concatenate(words) if words else np.zeros(0, np.int64))
        out = torch.zeros(sum(cnt_out_l), dtype=torch.int64, device=self.device)
        self.dist.all_to_all_single(out, inp, cnt_out_l, [w.shape[0] for w in words])
        self.bytes_sent += 8 * sum(w.shape[0] for i, w in enumerate(words) if i != self.rank)
        flat = out.cpu().numpy()
        res, o = [], 0
        for c in cnt_out_l:
            res.append(flat[o:o + c].view(dt).reshape(-1) if per else flat[o:o + c])
            o += c
        return res

    def all_reduce_sum(self, a):
        if self.dist is None:
            return a
        t = self._t(a.astype(np.int64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def all_gather_v(self, a):
        if self.dist is None:
            return a
        return np.concatenate(self.all_to_all_v([a.astype(np.int64)] * self.world))


def find_overlaps_and_create_piles_sharded(eng, rs_all: seqio.ReadSet, comm: Comm, freq=0.001, kmax=32,
                                           use_minhash=False, flush_bases=1 << 30, index_batch_bases=1 << 32):
    """Returns dict(lo, hi, pile_data, pile_off, overlaps, overlap_off, stats) for the reads [lo, hi) this rank
    owns; arrays are laid out like rvn_pass1_fetch_* restricted to that range.  A read set beyond index_batch_bases is
    indexed in several batches as the reference does (construct.cc:32-37): per batch the members' minimizers and — as
    query-only entries — the minhash-selected minimizers of every EARLIER read go to the hash owners, the shard is built
    and filtered, and every read up to the batch's end is mapped against it in flush windows."""
    from . import hip
    g, world = comm.rank, comm.world
    n_total = rs_all.n
    bounds = partition_reads(rs_all.lengths, world)
    lo, hi = int(bounds[g]), int(bounds[g + 1])
    own = eng.upload(slice_reads(rs_all, lo, hi))
    empty = np.zeros(0, dtype=hip.OVERLAP_DTYPE)
    p = eng.shard_piles_create(rs_all.lengths)
    n_matches_sent = n_overlaps_sent = n_map_overlaps = n_min_sent = 0
    occ = 0
    for b_first, b_last in index_batches(rs_all.lengths, index_batch_bases):
        # 1. sketch own reads: members of the batch, and the earlier reads as query-only entries; minimizers to the owner
        #    of their hash class (stable: keeps (read, position) order; the earlier reads come first, as their ids do)
        f_hi = min(max(b_first, lo), hi) - lo            # own reads [0, f_hi) lie before the batch
        m_lo, m_hi = f_hi, min(max(b_last, lo), hi) - lo  # own reads [m_lo, m_hi) are members
        pieces = []
        if f_hi > 0:
            pieces.append(eng.shard_sketch_fetch(eng.shard_sketch_range_count(own, 0, f_hi, use_minhash, foreign=True)))
        if m_hi > m_lo:
            pieces.append(eng.shard_sketch_fetch(eng.shard_sketch_range_count(own, m_lo, m_hi, use_minhash, foreign=False)))
        val = np.concatenate([x[0] for x in pieces]) if pieces else np.zeros(0, np.uint64)
        org = np.concatenate([x[1] for x in pieces]) if pieces else np.zeros(0, np.uint64)
        owner = hash_owner(val, world)
        order = np.argsort(owner, kind="stable")
        cnt = np.bincount(owner, minlength=world)
        cuts = np.concatenate([[0], np.cumsum(cnt)])
        val_s, org_s = val[order], org[order]
        val_r = comm.all_to_all_v([val_s[cuts[h]:cuts[h + 1]] for h in range(world)])
        org_r = comm.all_to_all_v([org_s[cuts[h]:cuts[h + 1]] for h in range(world)])
        n_min_sent += int(val.shape[0] - cnt[g])

        # 2. index shard of this hash class; 3. exact global Filter (keys = runs with members)
        eng.shard_index_build(np.concatenate(val_r), np.concatenate(org_r), all_query=use_minhash)
        counts = eng.shard_key_counts()
        occ = global_occurrence(counts[counts > 0], freq, comm)
        eng.set_occurrence(occ)

        # 4.-6. per flush window of query reads, exactly as the reference flushes (merge + AddLayers + truncation per window)
        for q_a, q_b in flush_windows(rs_all.lengths[:b_last], flush_bases):
            # self-join of the shard for the window's query reads; candidate pairs to the owner of the query read
            grp, pos, seg = eng.shard_join(n_total, True, True, q_a, q_b)
            per_read = np.diff(seg.astype(np.int64))
            m_cuts = [int(seg[bounds[h]]) for h in range(world)] + [int(seg[n_total])]
            cnt_r = comm.all_to_all_v([per_read[bounds[h]:bounds[h + 1]] for h in range(world)])
            grp_r = comm.all_to_all_v([grp[m_cuts[h]:m_cuts[h + 1]] for h in range(world)])
            pos_r = comm.all_to_all_v([pos[m_cuts[h]:m_cuts[h + 1]] for h in range(world)])
            seg_own, (grp_own, pos_own) = regroup_by_read(cnt_r, list(zip(grp_r, pos_r)))
            n_matches_sent += int(grp.shape[0] - (m_cuts[g + 1] - m_cuts[g]))
            # chain own reads' matches
            ovl, _ = eng.shard_chain(own, grp_own, pos_own, seg_own)
            n_map_overlaps += int(ovl.shape[0])
            # overlaps also to the owner of their rhs read (own ones are already here); merge + piles
            rhs_owner = np.searchsorted(bounds, ovl["rhs_id"].astype(np.int64), side="right") - 1
            recv = comm.all_to_all_v([ovl[rhs_owner == h] if h != g else empty for h in range(world)])
            n_overlaps_sent += int(np.sum(rhs_owner != g)) if ovl.shape[0] else 0
            for s_ in range(g + 1, world):
                assert recv[s_].shape[0] == 0, "avoid_symmetric: overlaps only travel to higher ranks"
            combined = np.concatenate([recv[s_] for s_ in range(g)] + [ovl]) if world > 1 else ovl
            p.merge(combined, kmax)
    data, poff = p.piles()
    kept, koff = p.overlaps()
    p.close()
    res = dict(lo=lo, hi=hi, occurrence=occ,
               pile_data=data[int(poff[lo]):int(poff[hi])].copy(),
               pile_off=(poff[lo:hi + 1] - poff[lo]).astype(np.uint64),
               overlaps=kept[int(koff[lo]):int(koff[hi])].copy(),
               overlap_off=(koff[lo:hi + 1] - koff[lo]).astype(np.uint32),
               stats=dict(minimizers_sent=n_min_sent, matches_sent=n_matches_sent,
                          overlaps_sent=n_overlaps_sent, map_overlaps=n_map_overlaps, bytes_sent=comm.bytes_sent))
    return res


# ---- device-resident variant: the exchange buffers are torch CUDA tensors, nothing crosses PCIe between stages ----
_MIX_I64 = 0x9E3779B97F4A7C15 - (1 << 64)


def hash_owner_t(values, world: int):
    """hash_owner on an int64 torch tensor (same bits as the numpy version: logical shift emulated by masking)."""
    import torch
    if world == 1:
        return torch.zeros_like(values)
    h = ((values * _MIX_I64) >> 33) & ((1 << 31) - 1)
    return h % world


def regroup_by_read_t(counts_per_src, data_per_src):
    """regroup_by_read on the device: per-source (per-read counts, tuple of flat int64 tensors)."""
    import torch
    total = torch.zeros_like(counts_per_src[0])
    for c in counts_per_src:
        total = total + c
    seg = torch.zeros(total.shape[0] + 1, dtype=torch.int64, device=total.device)
    torch.cumsum(total, 0, out=seg[1:])
    n_out = int(seg[-1].item())
    outs = [torch.empty(n_out, dtype=torch.int64, device=total.device) for _ in data_per_src[0]]
    start = seg[:-1].clone()
    for c, datas in zip(counts_per_src, data_per_src):
        m = int(c.sum().item())
        if m:
            src_off = torch.cumsum(c, 0) - c
            dest = torch.repeat_interleave(start - src_off, c) + torch.arange(m, dtype=torch.int64, device=c.device)
            for o, d in zip(outs, datas):
                o[dest] = d
        start = start + c
    return seg, outs


class DeviceComm(Comm):
    """Comm whose payloads are int64 torch tensors on the GPU (RCCL: backend "nccl")."""

    def all_to_all_t(self, parts):
        """parts[h]: 1-D int64 CUDA tensor for rank h.  Returns the list received, indexed by source rank."""
        import torch
        if self.dist is None:
            return [parts[0]]
        dev = parts[0].device
        cnt_in = torch.tensor([int(p.shape[0]) for p in parts], dtype=torch.int64, device=dev)
        cnt_out = torch.zeros(self.world, dtype=torch.int64, device=dev)
        self.dist.all_to_all_single(cnt_out, cnt_in)
        out_l = [int(x) for x in cnt_out.tolist()]
        inp = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=dev)
        out = torch.empty(sum(out_l), dtype=torch.int64, device=dev)
        self.dist.all_to_all_single(out, inp, out_l, [int(p.shape[0]) for p in parts])
        self.bytes_sent += 8 * sum(int(p.shape[0]) for i, p in enumerate(parts) if i != self.rank)
        return list(torch.split(out, out_l))


def _all_to_all_flat_t(self, flat, send_lens):
    """flat: 1-D int64 CUDA tensor holding the parts for ranks 0..world-1 back to back (send_lens elements each).
    Returns (flat receive tensor, receive lengths by source rank)."""
    import torch
    send_lens = [int(x) for x in send_lens]
    if self.dist is None:
        return flat[:send_lens[0]], [send_lens[0]]
    dev = flat.device
    cnt_in = torch.tensor(send_lens, dtype=torch.int64, device=dev)
    cnt_out = torch.zeros(self.world, dtype=torch.int64, device=dev)
    self.dist.all_to_all_single(cnt_out, cnt_in)
    out_l = [int(x) for x in cnt_out.tolist()]
    out = torch.empty(sum(out_l), dtype=torch.int64, device=dev)
    self.dist.all_to_all_single(out, flat[:sum(send_lens)].contiguous(), out_l, send_lens)
    self.bytes_sent += 8 * sum(c for i, c in enumerate(send_lens) if i != self.rank)
    return out, out_l


DeviceComm.all_to_all_flat_t = _all_to_all_flat_t


def find_overlaps_and_create_piles_sharded_dev(eng, rs_all: seqio.ReadSet, comm, device, freq=0.001, kmax=32,
                                               use_minhash=False, flush_bases=1 << 30, own=None, laps=None, fetch=True,
                                               index_batch_bases=1 << 32):
    """find_overlaps_and_create_piles_sharded with every exchange buffer resident in HBM (`device`: torch device of
    the engine's GPU; `comm`: DeviceComm or a test double with all_to_all_t / all_reduce_sum / all_gather_v).
    torch allocates the exchange buffers and carries the collectives; partitioning by owner, regrouping per read and
    the merge offsets are the engine's own kernels (raven_amd/csrc/shard.hip).  Several index batches as in the host
    variant (construct.cc:32-37)."""
    import torch
    g, world = comm.rank, comm.world
    n_total = rs_all.n
    bounds = partition_reads(rs_all.lengths, world)
    lo, hi = int(bounds[g]), int(bounds[g + 1])
    if own is None:  # `own`: this rank's reads already resident (uploaded once, as a caller running several passes does)
        own = eng.upload(slice_reads(rs_all, lo, hi))
    i64 = dict(dtype=torch.int64, device=device)
    import time as _time
    _t = [_time.perf_counter()]

    def lap(name):  # laps: optional dict collecting wall time per stage (synchronising: debugging / DESIGN.md numbers)
        if laps is not None:
            torch.cuda.synchronize(device)
            now = _time.perf_counter()
            laps[name] = laps.get(name, 0.0) + (now - _t[0])
            _t[0] = now

    def sync():
        # the engine works on its own stream: before it WRITES into freshly allocated torch memory (which the caching
        # allocator may have recycled from tensors with torch / RCCL work still in flight) and before it READS what a
        # collective produced, the device is synchronised
        torch.cuda.synchronize(device)

    b_list = [int(x) for x in bounds]
    r_split = [b_list[h + 1] - b_list[h] for h in range(world)]
    n_own = hi - lo
    p = eng.shard_piles_create(rs_all.lengths)
    n_matches_sent = n_sent = n_map = n_min_sent = 0
    occ = 0
    for b_first, b_last in index_batches(rs_all.lengths, index_batch_bases):
        # 1. sketch: the members of the batch and, as query-only entries, the minhash-selected minimizers of this rank's
        #    EARLIER reads; minimizers to the owner of their hash class (stable partition keeps (read, position) order)
        f_hi = min(max(b_first, lo), hi) - lo
        m_lo, m_hi = f_hi, min(max(b_last, lo), hi) - lo
        pieces = []
        for first, last, foreign in ((0, f_hi, True), (m_lo, m_hi, False)):
            if last > first:
                k = eng.shard_sketch_range_count(own, first, last, use_minhash, foreign=foreign)
                v, o = torch.empty(k, **i64), torch.empty(k, **i64)
                sync()
                if k:
                    eng.shard_sketch_fetch_dev(v.data_ptr(), o.data_ptr())
                pieces.append((v, o))
        if len(pieces) == 1:
            val, org = pieces[0]
        elif pieces:
            val, org = torch.cat([x[0] for x in pieces]), torch.cat([x[1] for x in pieces])
        else:
            val, org = torch.empty(0, **i64), torch.empty(0, **i64)
        del pieces
        n = int(val.shape[0])
        if world == 1:  # one owner: nothing to partition
            val_p, org_p, cnt = val, org, [n]
        else:
            val_p, org_p = torch.empty(n, **i64), torch.empty(n, **i64)
            sync()
            cnt = eng.shard_split_minimizers_dev(val.data_ptr(), org.data_ptr(), n, world, val_p.data_ptr(), org_p.data_ptr())
        del val, org
        vcat = comm.all_to_all_flat_t(val_p, cnt)[0]
        ocat = comm.all_to_all_flat_t(org_p, cnt)[0]
        del val_p, org_p
        n_min_sent += int(n - cnt[g])
        lap("sketch+split+exchange1")

        # 2. index shard; 3. exact global Filter
        sync()
        n_flagged = eng.shard_count_flagged_dev(ocat.data_ptr(), ocat.shape[0])
        eng.shard_index_build_dev(vcat.data_ptr(), ocat.data_ptr(), vcat.shape[0], use_minhash, n_flagged)
        hist, over = eng.shard_key_histogram()
        occ = occurrence_from_histogram(hist, over, freq, comm)
        eng.set_occurrence(occ)
        lap("index+filter")

        # 4.-6. per flush window of query reads (merge + AddLayers + truncation per window, as the reference flushes)
        for q_a, q_b in flush_windows(rs_all.lengths[:b_last], flush_bases):
            n_m = eng.shard_join_count(n_total, True, True, q_a, q_b)
            grp, pos = torch.empty(n_m, **i64), torch.empty(n_m, **i64)
            seg = torch.empty(n_total + 1, **i64)
            per_read = torch.empty(n_total, **i64)
            sync()
            eng.shard_join_fetch_dev(grp.data_ptr(), pos.data_ptr(), seg.data_ptr())
            eng.shard_adjacent_diff_dev(seg.data_ptr(), n_total, per_read.data_ptr())
            lap("join")
            m_cuts = seg[b_list].tolist()  # matches are in read order: the cut points of the read ranges
            m_split = [int(m_cuts[h + 1] - m_cuts[h]) for h in range(world)]
            cnt_flat, cnt_lens = comm.all_to_all_flat_t(per_read, r_split)
            grp_flat, m_lens = comm.all_to_all_flat_t(grp, m_split)
            pos_flat, _ = comm.all_to_all_flat_t(pos, m_split)
            n_in = sum(m_lens)
            seg_own = torch.empty(n_own + 1, **i64)
            grp_own, pos_own = torch.empty(n_in, **i64), torch.empty(n_in, **i64)
            sync()
            c_ptr, g_ptr, p_ptr, at_c, at_m = [], [], [], 0, 0
            for h in range(world):  # per-source views into the flat receive buffers
                c_ptr.append(cnt_flat.data_ptr() + 8 * at_c)
                g_ptr.append(grp_flat.data_ptr() + 8 * at_m)
                p_ptr.append(pos_flat.data_ptr() + 8 * at_m)
                at_c += cnt_lens[h]
                at_m += m_lens[h]
            eng.shard_regroup_dev(c_ptr, g_ptr, p_ptr, m_lens, n_own, seg_own.data_ptr(), grp_own.data_ptr(), pos_own.data_ptr())
            n_matches_sent += int(n_m - m_split[g])
            lap("exchange2+regroup")
            # chain
            n_o = eng.shard_chain_dev(own, grp_own.data_ptr(), pos_own.data_ptr(), seg_own.data_ptr(), n_in)
            ovl = torch.empty((n_o, 4), **i64)
            ovl_p = torch.empty((n_o, 4), **i64)
            off_own = torch.empty(own.n + 1, dtype=torch.int32, device=device)
            sync()
            eng.map_fetch_dev(ovl.data_ptr(), off_own.data_ptr())
            n_map += int(n_o)
            lap("chain")
            # overlaps also to the owner of their rhs read (stable partition; the own ones stay); merge + piles
            o_cnt = eng.shard_split_overlaps_dev(ovl.data_ptr(), n_o, b_list, world, g, ovl_p.data_ptr())
            send = [4 * c for c in o_cnt[:world]]
            n_sent += sum(o_cnt[:world])
            recv_flat, recv_lens = comm.all_to_all_flat_t(ovl_p.reshape(-1)[:sum(send)], send)
            for s_ in range(g + 1, world):
                assert recv_lens[s_] == 0, "avoid_symmetric: overlaps only travel to higher ranks"
            sync()
            parts, at = [], 0
            for s_ in range(world):
                if recv_lens[s_]:
                    parts.append((recv_flat.data_ptr() + 8 * at, recv_lens[s_] // 4))
                at += recv_lens[s_]
            parts.append((ovl.data_ptr(), n_o))
            p.merge_parts_dev(parts, kmax)
            lap("exchange3+merge")
    stats = dict(minimizers_sent=n_min_sent, matches_sent=n_matches_sent, overlaps_sent=int(n_sent),
                 map_overlaps=n_map, bytes_sent=comm.bytes_sent)
    if not fetch:  # the piles and overlap lists of this rank's reads stay in HBM: p.piles() / p.overlaps() / p.close()
        return dict(lo=lo, hi=hi, occurrence=occ, pass1=p, stats=stats)
    data, poff = p.piles()
    kept, koff = p.overlaps()
    p.close()
    lap("fetch piles+overlaps")
    return dict(lo=lo, hi=hi, occurrence=occ,
                pile_data=data[int(poff[lo]):int(poff[hi])].copy(),
                pile_off=(poff[lo:hi + 1] - poff[lo]).astype(np.uint64),
                overlaps=kept[int(koff[lo]):int(koff[hi])].copy(),
                overlap_off=(koff[lo:hi + 1] - koff[lo]).astype(np.uint32),
                stats=stats)


# ---- polishing round sharded by windows (windows are independent; no data-path collective but the final gather) ----
def polish_round_sharded(eng, targets, reads, comm, targets_rs, quals=None, q=0.0, err=0.3, w=500, trim=True,
                         m=3, n=-5, g=-4):
    """One racon round sharded over the ranks.  Every rank holds the targets and the read set (`targets`, `reads`:
    uploaded handles; `targets_rs`: the host ReadSet of the targets).
      1. reads are mapped independently of each other: rank g maps the reads [n g / N, n (g+1) / N) against the (replicated,
         cheap) target index and keeps the best overlap per read; the table (36 B per read) is all-gathered;
      2. windows are independent: rank g aligns the overlaps that touch its window range and runs the POA of those
         windows only; the per-target consensus pieces are all-gathered and concatenated in rank order.
    Both splits reproduce the single-GPU round byte for byte.  Returns (consensus list, ratio)."""
    lengths = targets_rs.lengths.astype(np.int64)
    n_win = int(((lengths + w - 1) // w).sum())
    lo = n_win * comm.rank // comm.world
    hi = n_win * (comm.rank + 1) // comm.world
    if comm.world > 1:
        n_reads = reads.n
        r_lo = n_reads * comm.rank // comm.world
        r_hi = n_reads * (comm.rank + 1) // comm.world
        best, bt, _ = eng.polish_map_best(targets, reads, r_lo, r_hi, err=err)
        packed = np.concatenate([best.astype(np.uint32), bt.reshape(-1, 1).astype(np.uint32),
                                 np.zeros((bt.shape[0], 1), np.uint32)], axis=1)  # 10 words per read -> 5 int64
        table = comm.all_gather_v(np.ascontiguousarray(packed).reshape(-1).view(np.int64)).view(np.uint32).reshape(-1, 10)
        assert table.shape[0] == n_reads
        eng.polish_set_best(table[:, :8], table[:, 8])
    cons, nw, npol, _ = eng.polish_round_range(targets, reads, lo, hi, quals=quals, q=q, err=err, w=w, trim=trim, m=m,
                                               n=n, g=g)
    nt = len(cons)
    lens = np.array([len(c) for c in cons], dtype=np.int64)
    all_lens = comm.all_gather_v(lens).reshape(comm.world, nt)
    flat = np.concatenate(cons) if nt else np.zeros(0, np.uint8)
    pad = (-flat.shape[0]) % 8
    words = np.concatenate([flat, np.zeros(pad, np.uint8)]).view(np.int64)
    word_cnt = comm.all_gather_v(np.array([words.shape[0]], dtype=np.int64))
    all_words = comm.all_gather_v(words)
    counts = comm.all_reduce_sum(np.stack([nw.astype(np.int64), npol.astype(np.int64)]))
    out = [[] for _ in range(nt)]
    wo = 0
    for r in range(comm.world):
        b = all_words[wo:wo + int(word_cnt[r])].view(np.uint8)
        wo += int(word_cnt[r])
        o = 0
        for t in range(nt):
            out[t].append(b[o:o + int(all_lens[r, t])])
            o += int(all_lens[r, t])
    ratio = np.where(counts[0] > 0, counts[1] / np.maximum(counts[0], 1), 0.0)
    return [np.concatenate(p) if p else np.zeros(0, np.uint8) for p in out], ratio
