"""ctypes binding of the CPU oracle (oracle/raven_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package `raven_amd` never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libraven_oracle.so")

OVERLAP_DTYPE = np.dtype([
    ("lhs_id", "<u4"), ("lhs_begin", "<u4"), ("lhs_end", "<u4"),
    ("rhs_id", "<u4"), ("rhs_begin", "<u4"), ("rhs_end", "<u4"),
    ("score", "<u4"), ("strand", "<u4")])


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("raven_oracle.cpp", "poa_oracle.cpp")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, u32, u64, i32, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
        L.orc_engine_create.restype = vp
        L.orc_engine_create.argtypes = [u32] * 6
        L.orc_engine_destroy.argtypes = [vp]
        L.orc_sketch.restype = u64
        L.orc_sketch.argtypes = [vp, vp, u32, u32, i32, vp, vp, u64]
        L.orc_engine_minimize.argtypes = [vp, vp, vp, vp, vp, u32, u32, i32, C.c_uint]
        L.orc_engine_filter.restype = i32
        L.orc_engine_filter.argtypes = [vp, dbl]
        L.orc_engine_occurrence.restype = u32
        L.orc_engine_occurrence.argtypes = [vp]
        L.orc_engine_find.restype = u32
        L.orc_engine_find.argtypes = [vp, u64, vp, u32]
        L.orc_engine_map.restype = u64
        L.orc_engine_map.argtypes = [vp, vp, u32, u32, i32, i32, i32, vp, u64, vp, u64, vp, vp, vp, u64, vp]
        L.orc_engine_counters.argtypes = [vp, vp]
        L.orc_chain.restype = u64
        L.orc_chain.argtypes = [vp, u32, vp, vp, u64, vp, u64]
        L.orc_pile_add_layers.argtypes = [vp, u32, vp, u64]
        L.orc_pile_trim_and_median.argtypes = [vp, u32, C.c_uint16, vp, vp, vp, vp]
        L.orc_truncate.restype = u64
        L.orc_truncate.argtypes = [vp, u64, u64]
        L.orc_find_overlaps_and_create_piles.restype = vp
        L.orc_find_overlaps_and_create_piles.argtypes = [vp, vp, vp, vp, vp, u32, dbl, u64, i32, u64, u64, C.c_uint]
        L.orc_pass1_destroy.argtypes = [vp]
        for name, rt in (("pile_words", u64), ("pile_data", vp), ("pile_offsets", vp), ("num_overlaps", u64),
                         ("overlaps", vp), ("overlap_offsets", vp), ("occurrence", u32), ("t_minimize", dbl),
                         ("t_map", dbl)):
            fn = getattr(L, "orc_pass1_" + name)
            fn.restype = rt
            fn.argtypes = [vp]
        L.orc_pile_add_kmers.argtypes = [vp, u32, vp, u64, u32, vp]
        L.orc_polish_round.restype = i32
        L.orc_polish_round.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp, u32, vp, vp, dbl, dbl, u32, i32, i32, i32, i32, vp, vp,
                                       vp, vp]
        L.orc_antiqsort.argtypes = [vp, u32]
        L.orc_poa_window.restype = i32
        L.orc_poa_window.argtypes = [vp, vp, vp, vp, vp, u32, i32, i32, i32, i32, vp, u32, C.POINTER(u32)]
        L.orc_poa_order_check.restype = i32
        L.orc_poa_order_check.argtypes = [vp, vp, vp, vp, u32, i32, i32, i32, vp]
        L.orc_poa_align_score_linear.restype = i32
        L.orc_poa_align_score_linear.argtypes = [vp, u32, vp, u32, i32, i32, i32]
        L.orc_edit_distance.restype = u32
        L.orc_edit_distance.argtypes = [C.c_char_p, u32, C.c_char_p, u32]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _copy(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


class Engine:
    """ram::MinimizerEngine restatement (defaults as in ram: bandwidth 500, chain 4, matches 100, gap 10000)."""

    def __init__(self, k=15, w=5, bandwidth=500, chain=4, matches=100, gap=10000):
        self.k, self.w = k, w
        self._h = lib().orc_engine_create(k, w, bandwidth, chain, matches, gap)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_engine_destroy(self._h)
            self._h = None

    def sketch(self, rs, i, minhash=False):
        L = int(rs.lengths[i])
        cap = L + 1
        v = np.zeros(cap, dtype=np.uint64)
        o = np.zeros(cap, dtype=np.uint64)
        words = rs.packed[int(rs.word_offsets[i]):]
        n = lib().orc_sketch(self._h, _p(words), L, int(rs.ids[i]), int(minhash), _p(v), _p(o), cap)
        return v[:n].copy(), o[:n].copy()

    def minimize(self, rs, first=0, last=None, minhash=False, threads=1):
        last = rs.n if last is None else last
        lib().orc_engine_minimize(self._h, _p(rs.packed), _p(rs.word_offsets), _p(rs.lengths), _p(rs.ids),
                                  first, last, int(minhash), threads)

    def filter(self, f):
        if lib().orc_engine_filter(self._h, float(f)) != 0:
            raise ValueError("[ram::MinimizerEngine::Filter] error: invalid frequency")

    @property
    def occurrence(self):
        return int(lib().orc_engine_occurrence(self._h))

    def set_occurrence(self, occurrence):
        L = lib()
        L.orc_engine_set_occurrence.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_engine_set_occurrence.restype = None
        L.orc_engine_set_occurrence(self._h, int(occurrence))

    def find(self, value, cap=1 << 16):
        o = np.zeros(cap, dtype=np.uint64)
        n = lib().orc_engine_find(self._h, int(value), _p(o), cap)
        return o[:min(n, cap)].copy(), n

    def map(self, rs, i, avoid_equal=True, avoid_symmetric=True, minhash=False, want_matches=False):
        L = int(rs.lengths[i])
        cap = 4096
        out = np.zeros(cap, dtype=OVERLAP_DTYPE)
        filt = np.zeros(L + 1, dtype=np.uint32)
        nf = C.c_uint64(0)
        nm = C.c_uint64(0)
        mcap = 1 << 22 if want_matches else 0
        mg = np.zeros(max(mcap, 1), dtype=np.uint64)
        mp = np.zeros(max(mcap, 1), dtype=np.uint64)
        words = rs.packed[int(rs.word_offsets[i]):]
        n = lib().orc_engine_map(self._h, _p(words), L, int(rs.ids[i]), int(avoid_equal), int(avoid_symmetric),
                                 int(minhash), _p(out), cap, _p(filt), L + 1, C.byref(nf),
                                 _p(mg) if want_matches else None, _p(mp) if want_matches else None, mcap,
                                 C.byref(nm))
        assert n <= cap
        res = dict(overlaps=out[:n].copy(), filtered=filt[:nf.value].copy(), n_matches=nm.value)
        if want_matches:
            assert nm.value <= mcap
            res["match_groups"] = mg[:nm.value].copy()
            res["match_positions"] = mp[:nm.value].copy()
        return res

    def counters(self):
        c = np.zeros(7, dtype=np.uint64)
        lib().orc_engine_counters(self._h, _p(c))
        return dict(zip(("index_bases", "index_minimizers", "index_keys", "query_bases", "query_minimizers",
                         "matches", "overlaps"), (int(x) for x in c)))

    def chain(self, lhs_id, groups, positions):
        groups = np.ascontiguousarray(groups, dtype=np.uint64)
        positions = np.ascontiguousarray(positions, dtype=np.uint64)
        cap = max(16, groups.shape[0])
        out = np.zeros(cap, dtype=OVERLAP_DTYPE)
        n = lib().orc_chain(self._h, lhs_id, _p(groups), _p(positions), groups.shape[0], _p(out), cap)
        return out[:n].copy()

    def find_overlaps_and_create_piles(self, rs, freq=0.001, kmax=32, use_minhash=False,
                                       index_batch_bases=1 << 32, flush_bases=1 << 30, threads=1):
        h = lib().orc_find_overlaps_and_create_piles(self._h, _p(rs.packed), _p(rs.word_offsets), _p(rs.lengths),
                                                     _p(rs.ids), rs.n, float(freq), kmax, int(use_minhash),
                                                     index_batch_bases, flush_bases, threads)
        L = lib()
        try:
            res = dict(
                pile_offsets=_copy(L.orc_pass1_pile_offsets(h), rs.n + 1, np.uint64),
                pile_data=_copy(L.orc_pass1_pile_data(h), L.orc_pass1_pile_words(h), np.uint16),
                overlap_offsets=_copy(L.orc_pass1_overlap_offsets(h), rs.n + 1, np.uint64),
                overlaps=_copy(L.orc_pass1_overlaps(h), L.orc_pass1_num_overlaps(h), OVERLAP_DTYPE),
                occurrence=int(L.orc_pass1_occurrence(h)),
                t_minimize=float(L.orc_pass1_t_minimize(h)),
                t_map=float(L.orc_pass1_t_map(h)),
                counters=self.counters())
        finally:
            L.orc_pass1_destroy(h)
        return res


def pile_add_layers(data: np.ndarray, pile_id: int, overlaps: np.ndarray) -> None:
    assert data.dtype == np.uint16 and overlaps.dtype == OVERLAP_DTYPE
    overlaps = np.ascontiguousarray(overlaps)
    lib().orc_pile_add_layers(_p(data), pile_id, _p(overlaps), overlaps.shape[0])


def truncate(overlaps: np.ndarray, kmax: int) -> np.ndarray:
    o = np.ascontiguousarray(overlaps).copy()
    n = lib().orc_truncate(_p(o), o.shape[0], kmax)
    return o[:n]


def pile_add_kmers(rs, i: int, positions: np.ndarray, kmer_len: int) -> np.ndarray:
    """Pile::AddKmers on read i: returns the kmers_ bitmap (cells + 1 entries, uint8)."""
    cells = int(rs.lengths[i]) >> 4
    out = np.zeros(cells + 1, dtype=np.uint8)
    positions = np.ascontiguousarray(positions, dtype=np.uint32)
    words = rs.packed[int(rs.word_offsets[i]):]
    lib().orc_pile_add_kmers(_p(words), int(rs.lengths[i]), _p(positions), positions.shape[0], kmer_len, _p(out))
    return out


def polish_round(targets, reads, quals=None, q=0.0, err=0.3, w=500, trim=True, m=3, n=-5, g=-4):
    """racon::Polisher::Polish restatement (one round). targets/reads: ReadSet; quals: list of Phred+33 arrays."""
    nt = targets.n
    ooff = np.zeros(nt + 1, dtype=np.uint64)
    np.cumsum(2 * targets.lengths.astype(np.uint64) + 1024, out=ooff[1:])
    out = np.zeros(int(ooff[-1]) + 1, dtype=np.uint8)
    out_len = np.zeros(nt, dtype=np.uint32)
    ratio = np.zeros(nt, dtype=np.float64)
    qa = qo = None
    if quals is not None:
        qo = np.zeros(len(quals) + 1, dtype=np.uint64)
        np.cumsum([len(x) for x in quals], out=qo[1:])
        qa = np.concatenate([np.asarray(x, dtype=np.uint8) for x in quals])
    rc = lib().orc_polish_round(_p(targets.packed), _p(targets.word_offsets), _p(targets.lengths), _p(targets.ids), nt,
                                _p(reads.packed), _p(reads.word_offsets), _p(reads.lengths), _p(reads.ids), reads.n,
                                _p(qa), _p(qo), float(q), float(err), w, int(trim), m, n, g, _p(out), _p(ooff),
                                _p(out_len), _p(ratio))
    assert rc == 0
    return [out[int(ooff[i]): int(ooff[i]) + int(out_len[i])].copy() for i in range(nt)], ratio


def find_chimeric_regions(data):
    """Pile::FindChimericRegions (pile.cc:176-187 with FindSlopes / MergeRegions) of one coverage array."""
    d = np.ascontiguousarray(data, dtype=np.uint16)
    out = np.zeros(max(2, d.shape[0]), dtype=np.uint32)
    L = lib()
    L.orc_find_chimeric_regions.restype = C.c_int64
    L.orc_find_chimeric_regions.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    n = L.orc_find_chimeric_regions(_p(d), d.shape[0], _p(out), out.shape[0] // 2)
    assert n >= 0
    return out[:2 * n].reshape(-1, 2).copy()


def overlap_update_and_type(overlaps, pile_begin, pile_end, pile_invalid):
    """OverlapUpdate + GetOverlapType (overlap_utils.cc:14-113) on a list: (updated overlaps, ok, type)."""
    o = np.ascontiguousarray(overlaps, dtype=OVERLAP_DTYPE).copy()
    b = np.ascontiguousarray(pile_begin, dtype=np.uint32)
    ok = np.zeros(o.shape[0], dtype=np.uint8)
    ty = np.zeros(o.shape[0], dtype=np.uint32)
    lib().orc_overlap_update_and_type(_p(o), C.c_uint64(o.shape[0]), _p(b), _p(np.ascontiguousarray(pile_end, dtype=np.uint32)),
                                      _p(np.ascontiguousarray(pile_invalid, dtype=np.uint8)), C.c_uint32(b.shape[0]), _p(ok),
                                      _p(ty))
    return o, ok, ty


def identity_filter(rs, overlaps, offsets, pile_begin, pile_end, pile_invalid, identity):
    """Identity filter loop of ResolveContainedReads (construct.cc:162-217) on per-pile lists (CSR)."""
    o = np.ascontiguousarray(overlaps, dtype=OVERLAP_DTYPE).copy()
    off = np.ascontiguousarray(offsets, dtype=np.uint32).copy()
    L = lib()
    L.orc_identity_filter.restype = C.c_uint64
    L.orc_identity_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_double]
    n = L.orc_identity_filter(_p(rs.packed), _p(rs.word_offsets), _p(rs.lengths), rs.n, _p(o), _p(off),
                              _p(np.ascontiguousarray(pile_begin, dtype=np.uint32)),
                              _p(np.ascontiguousarray(pile_end, dtype=np.uint32)),
                              _p(np.ascontiguousarray(pile_invalid, dtype=np.uint8)), float(identity))
    return o[:n], off


def second_pass(k, w, rs, pile_begin, pile_end, pile_invalid, freq=0.001, kmer_len=None, identity=0.0, batch_bases=1 << 30):
    """raven::FindOverlapsAndRepetetiveRegions (construct.cc:316-491) restated: dict(overlaps, contained, kmers)."""
    eng = Engine(k, w)
    n = rs.n
    inv = np.ascontiguousarray(pile_invalid, dtype=np.uint8)
    koff = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum((rs.lengths.astype(np.uint64) >> np.uint64(4)) + np.uint64(1), out=koff[1:])
    kmers = np.zeros(int(koff[-1]), dtype=np.uint8)
    contained = np.zeros(n, dtype=np.uint8)
    cap = 64 * n + 1024
    out = np.zeros(cap, dtype=OVERLAP_DTYPE)
    L = lib()
    L.orc_second_pass.restype = C.c_int64
    L.orc_second_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_double, C.c_uint32, C.c_double, C.c_uint64, C.c_void_p, C.c_uint64,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
    m = L.orc_second_pass(eng._h, _p(rs.packed), _p(rs.word_offsets), _p(rs.lengths), n,
                          _p(np.ascontiguousarray(pile_begin, dtype=np.uint32)),
                          _p(np.ascontiguousarray(pile_end, dtype=np.uint32)), _p(inv), float(freq),
                          int(k if kmer_len is None else kmer_len), float(identity), int(batch_bases), _p(out), cap,
                          _p(contained), _p(kmers), _p(koff))
    assert m >= 0
    return dict(overlaps=out[:m].copy(), contained=contained,
                kmers=[kmers[int(koff[i]):int(koff[i + 1])] if not inv[i] else kmers[:0] for i in range(n)])


def polish_layers(targets, reads, quals=None, q=0.0, err=0.3, w=500):
    """Steps 1-4 of racon's round (map, best overlap, NW path, breakpoints, layer rules): uint32[n, 7] rows
    {window, read, first base in the oriented read, bases, begin, end, rc}, in racon's order."""
    qa = qo = None
    if quals is not None:
        qo = np.zeros(len(quals) + 1, dtype=np.uint64)
        np.cumsum([len(x) for x in quals], out=qo[1:])
        qa = np.concatenate([np.asarray(x, dtype=np.uint8) for x in quals])
    L = lib()
    L.orc_polish_layers.restype = C.c_uint64
    L.orc_polish_layers.argtypes = [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_double, C.c_double, C.c_uint32, C.c_void_p, C.c_uint64]
    cap = int(reads.n) * (int(targets.lengths.max()) // w + 2) + 16
    cap = min(cap, int(sum((int(x) // w + 2) for x in reads.lengths)) + 16)
    out = np.zeros((cap, 7), dtype=np.uint32)
    n = L.orc_polish_layers(_p(targets.packed), _p(targets.word_offsets), _p(targets.lengths), _p(targets.ids), targets.n,
                            _p(reads.packed), _p(reads.word_offsets), _p(reads.lengths), _p(reads.ids), reads.n,
                            _p(qa), _p(qo), float(q), float(err), w, _p(out), cap)
    assert n <= cap
    return out[:n].copy()


def nw_full_matrix(on: bool):
    """The alignment paths of nw_breakpoints / polish_layers / polish_round from the full DP matrix (True) instead of the
    Ukkonen band that gives the same path (default; tests/test_oracle.py compares the two)."""
    lib().orc_nw_full(int(bool(on)))


def nw_breakpoints(query: np.ndarray, target: np.ndarray, q_begin: int, t_begin: int, w: int):
    """Global alignment path of two code arrays (plain DP, ties: diagonal, then query base only, then target base
    only) -> racon's find_breaking_points_from_cigar.  Returns (pairs[(t, q)] two per window with an aligned pair,
    edit distance)."""
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    cap = 4 * ((t_begin + len(target)) // w + 2)
    out = np.zeros(cap, dtype=np.uint32)
    dist = np.zeros(1, dtype=np.uint32)
    L = lib()
    L.orc_nw_breakpoints.restype = C.c_uint64
    L.orc_nw_breakpoints.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_void_p, C.c_uint64, C.c_void_p]
    n = L.orc_nw_breakpoints(_p(query), len(query), _p(target), len(target), q_begin, t_begin, w, _p(out), cap, _p(dist))
    return out[: 2 * n].reshape(-1, 2).copy(), int(dist[0])


def antiqsort(n: int) -> np.ndarray:
    """Values (ascending-sort killer) that drive libstdc++'s introsort into its heapsort fallback."""
    out = np.zeros(n, dtype=np.uint32)
    lib().orc_antiqsort(_p(out), n)
    return out


def poa_window(layers, begins=None, ends=None, quals=None, m=3, n=-5, g=-4, trim=True, device_order=False, end_tie=0,
               order_where=0):
    """racon Window::GenerateConsensus: layers = list of uint8 code arrays (layers[0] = backbone), begins/ends =
    backbone positions of each layer (ignored for the backbone), quals = list of uint8 Phred+33 arrays or None.
    device_order: the rows of the graph in the device kernels' incremental order instead of spoa's DFS rank (another
    valid topological order: only ties between equal scores can come out differently).
    end_tie: the end node of a layer's alignment among equal scores: 0 = spoa's rule (first in rank order), 1 = smallest
    node id (the device kernels' rule), 2 = largest.  device_order=True, end_tie=1 is the statement of what the device
    kernels compute.  order_where (diagnostic): 1 / 2 = the device's order in the alignments / in the consensus only.
    Both settings apply to this call only.
    Returns (consensus codes, polished flag)."""
    k = len(layers)
    off = np.zeros(k + 1, dtype=np.uint64)
    np.cumsum([len(x) for x in layers], out=off[1:])
    codes = np.concatenate([np.asarray(x, dtype=np.uint8) for x in layers]) if k else np.zeros(0, np.uint8)
    q = None
    if quals is not None:
        q = np.concatenate([np.asarray(x, dtype=np.uint8) for x in quals])
    blen = len(layers[0])
    b = np.asarray([0] * k if begins is None else begins, dtype=np.uint32)
    e = np.asarray([blen - 1] * k if ends is None else ends, dtype=np.uint32)
    cap = int(off[-1]) + 16
    out = np.zeros(cap, dtype=np.uint8)
    n_out = C.c_uint32(0)
    polished = lib().orc_poa_window(_p(codes), _p(q), _p(off), _p(b), _p(e), k, m, n, g, int(trim) | (2 if device_order else 0) | ((end_tie & 3) << 2) | ((order_where & 3) << 4), _p(out), cap,
                                    C.byref(n_out))
    if polished < 0:
        raise ValueError("[racon::Window::AddLayer] error: layer begin and end positions are invalid!")
    return out[:n_out.value].copy(), bool(polished)


def poa_order_check(layers, begins=None, ends=None, m=3, n=-5, g=-4):
    """Replays the device kernel's incremental topological-order rule beside spoa's graph construction;
    returns -1 if every edge keeps rank(tail) < rank(head) after every layer, else the failing layer."""
    k = len(layers)
    off = np.zeros(k + 1, dtype=np.uint64)
    np.cumsum([len(x) for x in layers], out=off[1:])
    codes = np.concatenate([np.asarray(x, dtype=np.uint8) for x in layers])
    blen = len(layers[0])
    b = np.asarray([0] * k if begins is None else begins, dtype=np.uint32)
    e = np.asarray([blen - 1] * k if ends is None else ends, dtype=np.uint32)
    info = np.zeros(8, dtype=np.int64)
    return int(lib().orc_poa_order_check(_p(codes), _p(off), _p(b), _p(e), k, m, n, g, _p(info))), info


def poa_align_score_linear(target, query, m=3, n=-5, g=-4) -> int:
    t = np.ascontiguousarray(target, dtype=np.uint8)
    q = np.ascontiguousarray(query, dtype=np.uint8)
    return int(lib().orc_poa_align_score_linear(_p(t), t.shape[0], _p(q), q.shape[0], m, n, g))


def edit_distance(a: bytes, b: bytes) -> int:
    return int(lib().orc_edit_distance(a, len(a), b, len(b)))


def pile_trim_and_median(data, coverage=4):
    """raven::Pile::FindValidRegion(coverage) + FindMedian on one pile (uint16 cells, modified in place as the reference
    does).  Returns (begin, end, median, invalid) in cell units."""
    assert data.dtype == np.uint16 and data.flags["C_CONTIGUOUS"]
    b, e = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
    m, inv = np.zeros(1, np.uint16), np.zeros(1, np.uint8)
    lib().orc_pile_trim_and_median(_p(data), data.shape[0], coverage, _p(b), _p(e), _p(m), _p(inv))
    return int(b[0]), int(e[0]), int(m[0]), bool(inv[0])
