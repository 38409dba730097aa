"""ctypes binding of libraven_hip.so (the C ABI in include/raven_hip.h).

This is the product path: it fails loudly when the HIP extension is missing or
no GPU is present — there is no CPU fallback (the CPU oracle lives in oracle/
and is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RVN_LIB_PATH") or os.path.join(_HERE, "lib", "libraven_hip.so")  # override: A/B builds
TEST_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libraven_hip_test.so")  # test hooks + host emulator, never the product

OVERLAP_DTYPE = np.dtype([
    ("lhs_id", "<u4"), ("lhs_begin", "<u4"), ("lhs_end", "<u4"),
    ("rhs_id", "<u4"), ("rhs_begin", "<u4"), ("rhs_end", "<u4"),
    ("score", "<u4"), ("strand", "<u4")])

ED_PAIR_DTYPE = np.dtype([
    ("lhs_read", "<u4"), ("lhs_begin", "<u4"), ("lhs_len", "<u4"),
    ("rhs_read", "<u4"), ("rhs_begin", "<u4"), ("rhs_len", "<u4"),
    ("strand", "<u4"), ("reserved", "<u4")])

RVN_OK, RVN_EINVAL, RVN_ENODEVICE, RVN_EHIP, RVN_ENOMEM = 0, -1, -2, -3, -4

# every symbol include/raven_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "rvn_group_create", "rvn_group_destroy", "rvn_group_size", "rvn_group_engine",
    "rvn_group_find_overlaps_and_create_piles", "rvn_group_polish_round",
    "rvn_last_error", "rvn_device_count", "rvn_engine_create", "rvn_engine_destroy", "rvn_reads_upload",
    "rvn_reads_destroy", "rvn_engine_minimize", "rvn_engine_filter", "rvn_engine_occurrence",
    "rvn_engine_map_batch", "rvn_engine_map_fetch", "rvn_engine_map_fetch_filtered",
    "rvn_find_overlaps_and_create_piles", "rvn_pass1_pile_words", "rvn_pass1_num_overlaps",
    "rvn_pass1_fetch_piles", "rvn_pass1_fetch_overlaps", "rvn_pass1_destroy", "rvn_pile_add_layers",
    "rvn_edit_distance_batch", "rvn_poa_consensus_batch", "rvn_pass1_trim_and_annotate", "rvn_poa_phase_cycles", "rvn_poa_narrow_windows", "rvn_polish_target_reads", "rvn_polish_set_chunk_windows", "rvn_polish_round_range", "rvn_polish_map_best", "rvn_polish_set_best", "rvn_shard_sketch", "rvn_shard_sketch_fetch",
    "rvn_shard_index_build", "rvn_shard_key_counts", "rvn_engine_set_occurrence", "rvn_shard_join",
    "rvn_shard_join_fetch", "rvn_shard_chain", "rvn_shard_piles", "rvn_shard_join_range", "rvn_shard_piles_create",
    "rvn_shard_piles_merge", "rvn_shard_piles_merge_dev", "rvn_shard_sketch_fetch_dev",
    "rvn_shard_index_build_dev", "rvn_shard_key_histogram", "rvn_shard_join_fetch_dev", "rvn_shard_chain_dev",
    "rvn_shard_split_minimizers_dev", "rvn_shard_count_flagged_dev", "rvn_shard_adjacent_diff_dev", "rvn_shard_regroup_dev",
    "rvn_shard_split_overlaps_dev", "rvn_shard_piles_merge_parts_dev",
    "rvn_engine_map_fetch_dev", "rvn_shard_piles_dev", "rvn_poa_set_mode", "rvn_poa_fallback_windows", "rvn_poa_wide_windows", "rvn_pile_add_kmers_batch",
    "rvn_reads_attach_quality", "rvn_polish_fetch_layers", "rvn_poa_work", "rvn_reads_upload_codes", "rvn_polish_round",
    "rvn_engine_sketch", "rvn_engine_sketch_fetch", "rvn_engine_index_size", "rvn_engine_index_fetch",
    "rvn_engine_counters", "rvn_engine_num_stages", "rvn_engine_stage_name", "rvn_engine_stage_ms",
    "rvn_engine_reset_stats", "rvn_engine_set_timing", "rvn_engine_set_kernel_timing",
    "rvn_engine_num_kernel_sites", "rvn_engine_kernel_site_name", "rvn_engine_kernel_ms",
    "rvn_engine_map_collect", "rvn_free",
    "rvn_find_overlaps_and_repetitive_regions", "rvn_pass2_num_overlaps", "rvn_pass2_kmer_cells", "rvn_pass2_fetch",
    "rvn_pass2_destroy", "rvn_engine_release_scratch", "rvn_filter_overlaps_by_identity", "rvn_pass1_find_chimeric_regions",
    "rvn_reads_load", "rvn_reads_name", "rvn_reads_info", "rvn_reads_fetch", "rvn_engine_set_option", "rvn_overlap_update_and_type", "rvn_group_polish_round_q", "rvn_group_peer_access", "rvn_shard_sketch_range", "rvn_group_find_overlaps_and_create_piles_batched",
    "rvn_polish_output_as_reads",
]

# TEST INFRASTRUCTURE: what include/raven_hip_test.h declares on top (libraven_hip_test.so only)
TEST_SYMBOLS = [
    "rvn_poa_banded_emulate", "rvn_test_low_complexity", "rvn_test_nw_breakpoints", "rvn_test_hash",
    "rvn_test_canonical", "rvn_test_std_sort_lendesc", "rvn_test_heap_sort_lendesc", "rvn_test_overlap_update_and_type", "rvn_test_find_chimeric_regions",
    "rvn_test_parse_file", "rvn_test_freelist", "rvn_test_inflate_fast",
]


class RavenHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libraven_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RavenHipError(
            "libraven_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or raven_amd/csrc/build.sh" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    _declare(L)
    _lib = L
    return L


_test_lib = None


def test_lib():
    """TEST INFRASTRUCTURE: libraven_hip_test.so = the product's objects + the rvn_test_* hooks and the host wavefront
    emulator (include/raven_hip_test.h).  Only tests/ and tools/ call this; the product binding above never does."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    if not os.path.exists(TEST_LIB_PATH):
        raise RavenHipError("libraven_hip_test.so not built (%s): run raven_amd/csrc/build.sh" % TEST_LIB_PATH)
    L = C.CDLL(TEST_LIB_PATH)
    _declare(L)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.rvn_test_low_complexity.restype = i32
    L.rvn_test_low_complexity.argtypes = [vp, u32]
    L.rvn_test_nw_breakpoints.argtypes = [vp, u32, vp, u32, u32, u32, u32, u32, i32, u32, u32, i32, vp, vp, vp]
    L.rvn_test_nw_breakpoints.restype = i32
    L.rvn_test_find_chimeric_regions.argtypes = [vp, u32, vp, u64]
    L.rvn_test_find_chimeric_regions.restype = C.c_int64
    L.rvn_test_overlap_update_and_type.argtypes = [vp, u64, vp, vp, vp, u32, vp, vp]
    L.rvn_poa_banded_emulate.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, i32, i32, i32, i32, vp, vp, vp, vp, i32]
    L.rvn_test_hash.restype = u64
    L.rvn_test_hash.argtypes = [u64, u32, i32]
    L.rvn_test_canonical.restype = i32
    L.rvn_test_canonical.argtypes = [vp, u32, u32, i32, C.POINTER(u64), C.POINTER(u32)]
    L.rvn_test_std_sort_lendesc.argtypes = [vp, u64]
    L.rvn_test_heap_sort_lendesc.argtypes = [vp, u64]
    pp = C.POINTER(C.c_void_p)
    L.rvn_test_parse_file.argtypes = [C.c_char_p, i32, u32, i32, u64, pp, pp, pp, C.POINTER(u32), pp, vp]
    L.rvn_test_freelist.argtypes = [u64, u64, vp, u32, vp, vp]
    L.rvn_test_inflate_fast.argtypes = [vp, u64, vp, u64, u64, vp]
    _test_lib = L
    return L


def _declare(L):
    vp, u32, u64, i32, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
    pp = C.POINTER(C.c_void_p)
    L.rvn_last_error.restype = C.c_char_p
    L.rvn_device_count.restype = i32
    L.rvn_engine_create.argtypes = [pp, u32, u32, u32, u32, u32, u32, i32]
    L.rvn_engine_destroy.argtypes = [vp]
    L.rvn_reads_upload.argtypes = [vp, vp, u64, vp, vp, vp, u32, pp]
    L.rvn_reads_destroy.argtypes = [vp]
    L.rvn_engine_minimize.argtypes = [vp, vp, u32, u32, i32]
    L.rvn_engine_filter.argtypes = [vp, dbl]
    L.rvn_engine_occurrence.restype = u32
    L.rvn_engine_occurrence.argtypes = [vp]
    L.rvn_engine_map_batch.argtypes = [vp, vp, u32, u32, i32, i32, i32, i32, C.POINTER(u64)]
    L.rvn_engine_map_fetch.argtypes = [vp, vp, vp]
    L.rvn_engine_map_fetch_filtered.argtypes = [vp, vp, vp, C.POINTER(u64)]
    L.rvn_engine_map_collect.argtypes = [vp, vp, u32, u32, i32, i32, i32, i32, pp, pp, pp, pp]
    L.rvn_free.argtypes = [vp]
    L.rvn_find_overlaps_and_create_piles.argtypes = [vp, vp, dbl, u32, i32, u64, u64, pp]
    L.rvn_pass1_pile_words.restype = u64
    L.rvn_pass1_pile_words.argtypes = [vp]
    L.rvn_pass1_num_overlaps.restype = u64
    L.rvn_pass1_num_overlaps.argtypes = [vp]
    L.rvn_pass1_fetch_piles.argtypes = [vp, vp, vp]
    L.rvn_pass1_fetch_overlaps.argtypes = [vp, vp, vp]
    L.rvn_pass1_destroy.argtypes = [vp]
    L.rvn_pile_add_layers.argtypes = [vp, vp, u32, u32, vp, u64]
    L.rvn_pile_add_kmers_batch.argtypes = [vp, vp, u32, u32, vp, vp, vp, vp]
    L.rvn_reads_attach_quality.argtypes = [vp, vp, vp, vp, i32]
    L.rvn_reads_upload_codes.argtypes = [vp, vp, vp, vp, u32, pp]
    L.rvn_polish_output_as_reads.argtypes = [vp, pp]
    L.rvn_find_overlaps_and_repetitive_regions.argtypes = [vp, vp, vp, vp, vp, dbl, u32, dbl, u64, pp]
    L.rvn_pass2_num_overlaps.argtypes = [vp]
    L.rvn_pass2_num_overlaps.restype = u64
    L.rvn_pass2_kmer_cells.argtypes = [vp]
    L.rvn_pass2_kmer_cells.restype = u64
    L.rvn_pass2_fetch.argtypes = [vp, vp, vp, vp, vp]
    L.rvn_pass2_destroy.argtypes = [vp]
    L.rvn_engine_release_scratch.argtypes = [vp]
    L.rvn_filter_overlaps_by_identity.argtypes = [vp, vp, vp, vp, vp, vp, vp, dbl]
    L.rvn_reads_load.argtypes = [vp, C.c_char_p, pp, vp]
    L.rvn_reads_name.argtypes = [vp, u32]
    L.rvn_reads_name.restype = C.c_char_p
    L.rvn_reads_info.argtypes = [vp, vp, vp, vp, vp, vp]
    L.rvn_reads_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
    L.rvn_pass1_find_chimeric_regions.argtypes = [vp, vp, vp, pp]
    L.rvn_poa_work.argtypes = [vp, vp]
    L.rvn_poa_work.restype = None
    L.rvn_polish_fetch_layers.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.rvn_polish_round.argtypes = [vp, vp, vp, vp, vp, dbl, dbl, u32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.rvn_edit_distance_batch.argtypes = [vp, vp, vp, u32, vp, C.POINTER(dbl), C.POINTER(u64)]
    L.rvn_poa_consensus_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, u32, i32, i32, i32, i32, vp, vp, vp, vp,
                                          C.POINTER(dbl)]
    L.rvn_poa_phase_cycles.argtypes = [vp, vp]
    L.rvn_polish_target_reads.argtypes = [vp, vp, u32]
    L.rvn_polish_set_chunk_windows.argtypes = [vp, u64]
    L.rvn_polish_set_chunk_windows.restype = u64
    L.rvn_shard_sketch.argtypes = [vp, vp, i32, C.POINTER(u64)]
    L.rvn_shard_sketch_fetch.argtypes = [vp, vp, vp]
    L.rvn_shard_index_build.argtypes = [vp, vp, vp, u64, i32]
    L.rvn_shard_key_counts.argtypes = [vp, vp]
    L.rvn_engine_set_occurrence.argtypes = [vp, u32]
    L.rvn_shard_join.argtypes = [vp, u32, i32, i32, C.POINTER(u64)]
    L.rvn_shard_join_fetch.argtypes = [vp, vp, vp, vp]
    L.rvn_shard_join_range.argtypes = [vp, u32, i32, i32, u32, u32, C.POINTER(u64)]
    L.rvn_shard_piles_create.argtypes = [vp, vp, u32, C.POINTER(vp)]
    L.rvn_shard_piles_merge.argtypes = [vp, vp, u64, u32]
    L.rvn_shard_piles_merge_dev.argtypes = [vp, vp, vp, u64, u32]
    L.rvn_shard_chain.argtypes = [vp, vp, vp, vp, vp, C.POINTER(u64)]
    L.rvn_shard_piles.argtypes = [vp, vp, u32, vp, u64, u32, C.POINTER(vp)]
    L.rvn_shard_sketch_fetch_dev.argtypes = [vp, vp, vp]
    L.rvn_shard_index_build_dev.argtypes = [vp, vp, vp, u64, i32, u64]
    L.rvn_shard_key_histogram.argtypes = [vp, vp, vp, u32, C.POINTER(u32)]
    L.rvn_shard_join_fetch_dev.argtypes = [vp, vp, vp, vp]
    L.rvn_shard_chain_dev.argtypes = [vp, vp, vp, vp, vp, u64, C.POINTER(u64)]
    L.rvn_engine_map_fetch_dev.argtypes = [vp, vp, vp]
    L.rvn_shard_piles_dev.argtypes = [vp, vp, u32, vp, vp, u64, u32, C.POINTER(vp)]
    L.rvn_poa_set_mode.argtypes = [vp, i32]
    L.rvn_poa_set_mode.restype = i32
    L.rvn_poa_fallback_windows.argtypes = [vp]
    L.rvn_poa_fallback_windows.restype = u32
    L.rvn_poa_wide_windows.argtypes = [vp]
    L.rvn_poa_wide_windows.restype = u32
    L.rvn_poa_narrow_windows.argtypes = [vp]
    L.rvn_poa_narrow_windows.restype = u32
    L.rvn_engine_sketch.argtypes = [vp, vp, u32, u32, i32, C.POINTER(u64)]
    L.rvn_engine_sketch_fetch.argtypes = [vp, vp, vp, vp]
    L.rvn_engine_index_size.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.rvn_engine_index_fetch.argtypes = [vp, vp, vp]
    L.rvn_engine_counters.argtypes = [vp, vp]
    L.rvn_engine_num_stages.restype = i32
    L.rvn_engine_stage_name.restype = C.c_char_p
    L.rvn_engine_stage_name.argtypes = [i32]
    L.rvn_engine_stage_ms.argtypes = [vp, vp, vp, i32]
    L.rvn_engine_reset_stats.argtypes = [vp]
    L.rvn_engine_set_timing.argtypes = [vp, i32]
    L.rvn_engine_set_kernel_timing.argtypes = [vp, i32]
    L.rvn_engine_num_kernel_sites.restype = i32
    L.rvn_engine_kernel_site_name.restype = C.c_char_p
    L.rvn_engine_kernel_site_name.argtypes = [i32]
    L.rvn_engine_kernel_ms.argtypes = [vp, vp, vp, i32]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _check(rc):
    if rc != RVN_OK:
        msg = lib().rvn_last_error().decode(errors="replace")
        if rc == RVN_EINVAL:
            raise ValueError(msg)
        raise RavenHipError("rc=%d: %s" % (rc, msg))


def device_count() -> int:
    return int(lib().rvn_device_count())


class CodeSet:
    """Just enough of a ReadSet for a Reads handle made from one-byte codes (Engine.upload_codes)."""

    def __init__(self, lengths):
        self.lengths = np.asarray(lengths, dtype=np.uint32)
        self.n = int(self.lengths.shape[0])
        self.ids = np.arange(self.n, dtype=np.uint32)

    @property
    def total_bases(self):
        return int(self.lengths.astype(np.int64).sum())


class Reads:
    def __init__(self, engine: "Engine", rs, codes=None, handle=None):
        self.rs = rs
        self.engine = engine
        h = C.c_void_p()
        if handle is not None:  # a read set the library made on the device (Engine.polish_output_as_reads)
            self._h = handle
            return
        if codes is not None:  # one-byte codes, packed on the device
            off = np.zeros(rs.n + 1, dtype=np.uint64)
            np.cumsum(rs.lengths.astype(np.uint64), out=off[1:])
            flat = np.ascontiguousarray(codes, dtype=np.uint8)
            assert flat.shape[0] == int(off[-1])
            _check(lib().rvn_reads_upload_codes(engine._h, _p(flat), _p(off), None, rs.n, C.byref(h)))
            self._h = h
            return
        packed = np.ascontiguousarray(rs.packed, dtype=np.uint64)
        n_words = int(rs.word_offsets[-1])
        _check(lib().rvn_reads_upload(engine._h, _p(packed), n_words,
                                      _p(np.ascontiguousarray(rs.word_offsets, dtype=np.uint64)),
                                      _p(np.ascontiguousarray(rs.lengths, dtype=np.uint32)),
                                      _p(np.ascontiguousarray(rs.ids, dtype=np.uint32)), rs.n, C.byref(h)))
        self._h = h

    @property
    def n(self):
        return self.rs.n

    def fetch(self):
        """Device-resident read set back on the host: (packed words, word offsets, lengths, quality bytes or None,
        quality offsets or None, quality shift)."""
        n, nw, nb, nq, sh = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        _check(lib().rvn_reads_info(self._h, C.byref(n), C.byref(nw), C.byref(nb), C.byref(nq), C.byref(sh)))
        packed = np.zeros(nw.value, dtype=np.uint64)
        woff = np.zeros(n.value + 1, dtype=np.uint64)
        lens = np.zeros(n.value, dtype=np.uint32)
        q = np.zeros(nq.value, dtype=np.uint8) if sh.value >= 0 else None
        qoff = np.zeros(n.value + 1, dtype=np.uint64) if sh.value >= 0 else None
        _check(lib().rvn_reads_fetch(self._h, _p(packed), _p(woff), _p(lens), _p(q), _p(qoff)))
        return packed, woff, lens, q, qoff, sh.value

    def attach_quality(self, quals, block_shift=0):
        """Keep the reads' qualities in HBM for the polishing rounds: `quals` = list of per-read uint8 arrays of
        Phred+33 bytes, one per 2^block_shift bases (0: per base; 6: biosoup block qualities + 33)."""
        if isinstance(quals, tuple):  # (flat uint8 array, uint64 offsets[n + 1]) as they are
            flat = np.ascontiguousarray(quals[0], dtype=np.uint8)
            off = np.ascontiguousarray(quals[1], dtype=np.uint64)
        else:
            lens = np.array([len(q) for q in quals], dtype=np.uint64)
            off = np.zeros(self.rs.n + 1, dtype=np.uint64)
            np.cumsum(lens, out=off[1:])
            flat = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.uint8) for q in quals])
                                        if len(quals) else np.zeros(0, np.uint8))
        _check(lib().rvn_reads_attach_quality(self.engine._h, self._h, _p(flat), _p(off), int(block_shift)))

    def close(self):
        if getattr(self, "_h", None):
            lib().rvn_reads_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass


class Pass1:
    def __init__(self, h, n):
        self._h = h
        self.n = n

    def piles(self):
        L = lib()
        words = int(L.rvn_pass1_pile_words(self._h))
        data = np.zeros(words, dtype=np.uint16)
        off = np.zeros(self.n + 1, dtype=np.uint64)
        _check(L.rvn_pass1_fetch_piles(self._h, _p(data), _p(off)))
        return data, off

    def merge(self, overlaps, kmax=32):
        """Sharded pass: one flush (merge + AddLayers + truncation) of overlaps in (query read, emission) order."""
        overlaps = np.ascontiguousarray(overlaps, dtype=OVERLAP_DTYPE)
        _check(lib().rvn_shard_piles_merge(self._h, _p(overlaps), overlaps.shape[0], kmax))

    def merge_parts_dev(self, parts, kmax=32):
        """rvn_shard_piles_merge_parts_dev: parts = [(device pointer, number of overlaps), ...] in ascending lhs order."""
        n = len(parts)
        ptrs = (C.c_void_p * max(n, 1))(*[int(a) for a, _ in parts])
        cnts = (C.c_uint64 * max(n, 1))(*[int(c) for _, c in parts])
        L = lib()
        L.rvn_shard_piles_merge_parts_dev.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        _check(L.rvn_shard_piles_merge_parts_dev(self._h, n, ptrs, cnts, kmax))

    def merge_dev(self, d_overlaps, d_read_off, n, kmax=32):
        _check(lib().rvn_shard_piles_merge_dev(self._h, d_overlaps, d_read_off, int(n), kmax))

    def trim_and_annotate(self, coverage=4):
        """Pile::FindValidRegion(coverage) + FindMedian for every pile, in place in HBM: (begin, end, median, invalid)."""
        L = lib()
        L.rvn_pass1_trim_and_annotate.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 4
        b = np.zeros(self.n, dtype=np.uint32)
        e = np.zeros(self.n, dtype=np.uint32)
        m = np.zeros(self.n, dtype=np.uint16)
        inv = np.zeros(self.n, dtype=np.uint8)
        _check(L.rvn_pass1_trim_and_annotate(self._h, int(coverage), _p(b), _p(e), _p(m), _p(inv)))
        return b, e, m, inv.astype(bool)

    def find_chimeric_regions(self, invalid):
        """Pile::FindChimericRegions of every valid pile (after trim_and_annotate): list of (k, 2) uint32 arrays of
        (begin, end) cells, one per pile."""
        inv = np.ascontiguousarray(invalid, dtype=np.uint8)
        off = np.zeros(self.n + 1, dtype=np.uint32)
        ptr = C.c_void_p()
        _check(lib().rvn_pass1_find_chimeric_regions(self._h, _p(inv), _p(off), C.byref(ptr)))
        total = int(off[-1])
        try:
            flat = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(max(2 * total, 1),))[:2 * total].copy()
        finally:
            lib().rvn_free(ptr)
        flat = flat.reshape(-1, 2)
        return [flat[int(off[i]):int(off[i + 1])] for i in range(self.n)]

    def overlaps(self):
        L = lib()
        n = int(L.rvn_pass1_num_overlaps(self._h))
        ovl = np.zeros(n, dtype=OVERLAP_DTYPE)
        off = np.zeros(self.n + 1, dtype=np.uint32)
        _check(L.rvn_pass1_fetch_overlaps(self._h, _p(ovl), _p(off)))
        return ovl, off

    def close(self):
        if getattr(self, "_h", None):
            lib().rvn_pass1_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass


def _pack_poa_windows(windows):
    """Flat arrays of a window batch as rvn_poa_consensus_batch / rvn_poa_banded_emulate take them."""
    codes, quals, loff, begins, ends, hasq, woff, ooff = [], [], [0], [], [], [], [0], [0]
    any_q = False
    for wdw in windows:
        layers = wdw["layers"]
        k = len(layers)
        blen = len(layers[0])
        b = wdw.get("begins") or [0] * k
        e_ = wdw.get("ends") or [max(blen - 1, 0)] * k
        q = wdw.get("quals")
        for i, lay in enumerate(layers):
            lay = np.asarray(lay, dtype=np.uint8)
            codes.append(lay)
            loff.append(loff[-1] + lay.shape[0])
            begins.append(int(b[i]))
            ends.append(int(e_[i]))
            if q is not None and q[i] is not None:
                quals.append(np.asarray(q[i], dtype=np.uint8))
                hasq.append(1)
                any_q = True
            else:
                quals.append(np.full(lay.shape[0], 33, dtype=np.uint8))
                hasq.append(0)
        woff.append(woff[-1] + k)
        ooff.append(ooff[-1] + 4 * blen + 256)
    nw = len(windows)
    return dict(codes=np.concatenate(codes) if codes else np.zeros(0, np.uint8),
                quals=np.concatenate(quals) if any_q else None, loff=np.asarray(loff, dtype=np.uint64),
                begins=np.asarray(begins, dtype=np.uint32), ends=np.asarray(ends, dtype=np.uint32),
                hasq=np.asarray(hasq, dtype=np.uint32), woff=np.asarray(woff, dtype=np.uint32), nw=nw,
                out=np.zeros(ooff[-1] + 16, dtype=np.uint8), ooff=np.asarray(ooff, dtype=np.uint64),
                out_len=np.zeros(nw, dtype=np.uint32), status=np.zeros(nw, dtype=np.uint32))


def _unpack_poa_consensus(a):
    ooff = a["ooff"]
    return [a["out"][int(ooff[i]): int(ooff[i]) + int(a["out_len"][i])].copy() for i in range(a["nw"])]


def poa_banded_emulate(windows, m=3, n=-5, g=-4, trim=True, variant=4):
    """TEST INFRASTRUCTURE: poa4.hip's phase functions stepped through on the HOST by the wavefront emulator (no GPU,
    no engine).  First attempt of the escalation chain only: status 8 = the window needs a wider band (or is beyond
    the kernel's limits).  Returns (list of consensus code arrays, status array)."""
    a = _pack_poa_windows(windows)
    T = test_lib()
    rc = T.rvn_poa_banded_emulate(
        _p(a["codes"]), _p(a["quals"]), _p(a["loff"]), _p(a["begins"]), _p(a["ends"]), _p(a["hasq"]), _p(a["woff"]),
        a["nw"], m, n, g, int(trim), _p(a["out"]), _p(a["ooff"]), _p(a["out_len"]), _p(a["status"]), int(variant))
    if rc != RVN_OK:
        raise (ValueError if rc == RVN_EINVAL else RavenHipError)(T.rvn_last_error().decode(errors="replace"))
    return _unpack_poa_consensus(a), a["status"]


class Engine:
    """Mirror of ram::MinimizerEngine over the C ABI (defaults as in ram)."""

    def __init__(self, k=15, w=5, bandwidth=500, chain=4, matches=100, gap=10000, device=0):
        self.k, self.w = min(max(k, 1), 31), w
        h = C.c_void_p()
        _check(lib().rvn_engine_create(C.byref(h), k, w, bandwidth, chain, matches, gap, device))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().rvn_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def upload(self, rs) -> Reads:
        return Reads(self, rs)

    def load(self, path) -> Reads:
        """raven::CreateParser(path) + Parse(-1) straight into HBM (rvn_reads_load): gzip members inflated by a pool of
        host threads, records found by one memchr pass, text cut into packed reads on the device.  The returned handle's .rs holds lengths / ids / names and .load_stats."""
        h = C.c_void_p()
        st = np.zeros(8, dtype=np.uint64)
        _check(lib().rvn_reads_load(self._h, str(path).encode(), C.byref(h), _p(st)))
        n = C.c_uint32(0)
        lib().rvn_reads_info(h, C.byref(n), None, None, None, None)
        lengths = np.zeros(n.value, dtype=np.uint32)
        lib().rvn_reads_fetch(h, None, None, _p(lengths), None, None)
        rs = CodeSet(lengths)
        rs.names = [lib().rvn_reads_name(h, i).decode() for i in range(n.value)]
        r = Reads.__new__(Reads)
        r.rs, r.engine, r._h = rs, self, h
        r.load_stats = {"n_sequences": int(st[0]), "n_bases": int(st[1]), "has_quality": int(st[2] & 0xFFFFFFFF),
                        "parse_s": float(st[3:4].view(np.float64)[0]), "device_s": float(st[4:5].view(np.float64)[0]),
                        "total_s": float(st[5:6].view(np.float64)[0]), "inflate_threads": int(st[6] & 0xFFFFFFFF),
                        "members": int(st[6] >> 32), "streaming": int(st[7] & 0xFFFFFFFF), "restarted": int(st[7] >> 32)}
        return r

    def upload_codes(self, code_arrays) -> Reads:
        """Read set from one-byte code arrays (values 0..3), packed on the device (rvn_reads_upload_codes): how the
        consensus of one polishing round becomes the target set of the next."""
        lens = [int(len(c)) for c in code_arrays]
        flat = np.concatenate([np.asarray(c, dtype=np.uint8) for c in code_arrays]) if lens else np.zeros(0, np.uint8)
        return Reads(self, CodeSet(lens), codes=flat)

    def polish_output_as_reads(self, lengths) -> Reads:
        """The consensus of this engine's last complete polish_round as the next round's target set, straight from HBM
        (rvn_polish_output_as_reads): the same read set upload_codes(<what the round returned>) gives, without the 100 MB
        of a C4 round going through host memory and PCIe again.  lengths = the lengths of the sequences the round returned."""
        h = C.c_void_p()
        _check(lib().rvn_polish_output_as_reads(self._h, C.byref(h)))
        return Reads(self, CodeSet([int(x) for x in lengths]), handle=h)

    # -- ram::MinimizerEngine interface -----------------------------------------------------
    def minimize(self, reads: Reads, first=0, last=None, minhash=False):
        last = reads.n if last is None else last
        _check(lib().rvn_engine_minimize(self._h, reads._h, first, last, int(minhash)))

    def filter(self, f):
        _check(lib().rvn_engine_filter(self._h, float(f)))

    @property
    def occurrence(self):
        return int(lib().rvn_engine_occurrence(self._h))

    def map_batch(self, reads: Reads, first=0, last=None, avoid_equal=True, avoid_symmetric=True, minhash=False,
                  want_filtered=False):
        last = reads.n if last is None else last
        n = C.c_uint64(0)
        _check(lib().rvn_engine_map_batch(self._h, reads._h, first, last, int(avoid_equal), int(avoid_symmetric),
                                          int(minhash), int(want_filtered), C.byref(n)))
        ovl = np.zeros(n.value, dtype=OVERLAP_DTYPE)
        off = np.zeros(last - first + 1, dtype=np.uint32)
        _check(lib().rvn_engine_map_fetch(self._h, _p(ovl), _p(off)))
        res = dict(overlaps=ovl, read_offsets=off)
        if want_filtered:
            tot = C.c_uint64(0)
            _check(lib().rvn_engine_map_fetch_filtered(self._h, None, None, C.byref(tot)))
            pos = np.zeros(tot.value, dtype=np.uint32)
            foff = np.zeros(last - first + 1, dtype=np.uint32)
            _check(lib().rvn_engine_map_fetch_filtered(self._h, _p(pos), _p(foff), C.byref(tot)))
            res["filtered"] = pos
            res["filtered_offsets"] = foff
        return res

    # -- raven::FindOverlapsAndCreatePiles ---------------------------------------------------
    def find_overlaps_and_create_piles(self, reads: Reads, freq=0.001, kmax=32, use_minhash=False,
                                       index_batch_bases=1 << 32, flush_bases=1 << 30) -> Pass1:
        h = C.c_void_p()
        _check(lib().rvn_find_overlaps_and_create_piles(self._h, reads._h, float(freq), kmax, int(use_minhash),
                                                        index_batch_bases, flush_bases, C.byref(h)))
        return Pass1(h, reads.n)

    # -- stage-level entry points of the sharded pass (raven_amd/sharded.py) ------------------------
    def shard_sketch(self, own_reads: Reads, index_minhash=False):
        n = C.c_uint64(0)
        _check(lib().rvn_shard_sketch(self._h, own_reads._h, int(index_minhash), C.byref(n)))
        values = np.zeros(n.value, dtype=np.uint64)
        origins = np.zeros(n.value, dtype=np.uint64)
        _check(lib().rvn_shard_sketch_fetch(self._h, _p(values), _p(origins)))
        return values, origins

    def shard_sketch_range_count(self, own_reads: Reads, first, last, index_minhash=False, foreign=False) -> int:
        """rvn_shard_sketch_range: reads [first, last) of the handle; foreign = reads of an earlier index batch (their
        minhash-selected minimizers as query-only entries).  Fetch with shard_sketch_fetch / shard_sketch_fetch_dev."""
        n = C.c_uint64(0)
        _check(lib().rvn_shard_sketch_range(self._h, own_reads._h, int(first), int(last), int(index_minhash), int(foreign),
                                            C.byref(n)))
        return int(n.value)

    def shard_sketch_fetch(self, n):
        values = np.zeros(n, dtype=np.uint64)
        origins = np.zeros(n, dtype=np.uint64)
        if n:
            _check(lib().rvn_shard_sketch_fetch(self._h, _p(values), _p(origins)))
        return values, origins

    def shard_index_build(self, values, origins, all_query=False):
        values = np.ascontiguousarray(values, dtype=np.uint64)
        origins = np.ascontiguousarray(origins, dtype=np.uint64)
        _check(lib().rvn_shard_index_build(self._h, _p(values), _p(origins), values.shape[0], int(all_query)))

    def shard_key_counts(self):
        m, u = C.c_uint64(0), C.c_uint64(0)
        _check(lib().rvn_engine_index_size(self._h, C.byref(m), C.byref(u)))
        counts = np.zeros(u.value, dtype=np.uint32)
        _check(lib().rvn_shard_key_counts(self._h, _p(counts)))
        return counts

    def set_occurrence(self, occurrence):
        _check(lib().rvn_engine_set_occurrence(self._h, int(occurrence)))

    def shard_join(self, n_reads_total, avoid_equal=True, avoid_symmetric=True, query_first=0, query_last=None):
        h = C.c_uint64(0)
        query_last = n_reads_total if query_last is None else query_last
        _check(lib().rvn_shard_join_range(self._h, n_reads_total, int(avoid_equal), int(avoid_symmetric), query_first,
                                          query_last, C.byref(h)))
        grp = np.zeros(h.value, dtype=np.uint64)
        pos = np.zeros(h.value, dtype=np.uint64)
        seg = np.zeros(n_reads_total + 1, dtype=np.uint64)
        _check(lib().rvn_shard_join_fetch(self._h, _p(grp), _p(pos), _p(seg)))
        return grp, pos, seg

    def shard_chain(self, own_reads: Reads, grp, pos, seg_off):
        grp = np.ascontiguousarray(grp, dtype=np.uint64)
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        seg_off = np.ascontiguousarray(seg_off, dtype=np.uint64)
        assert seg_off.shape[0] == own_reads.n + 1
        n = C.c_uint64(0)
        _check(lib().rvn_shard_chain(self._h, own_reads._h, _p(grp), _p(pos), _p(seg_off), C.byref(n)))
        ovl = np.zeros(n.value, dtype=OVERLAP_DTYPE)
        off = np.zeros(own_reads.n + 1, dtype=np.uint32)
        _check(lib().rvn_engine_map_fetch(self._h, _p(ovl), _p(off)))
        return ovl, off

    def shard_piles_create(self, lengths) -> Pass1:
        """Empty piles of ALL reads; Pass1.merge / merge_dev add the Map outputs of one flush window each."""
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        h = C.c_void_p()
        _check(lib().rvn_shard_piles_create(self._h, _p(lengths), lengths.shape[0], C.byref(h)))
        return Pass1(h, lengths.shape[0])

    def shard_piles(self, lengths, overlaps, kmax=32) -> Pass1:
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        overlaps = np.ascontiguousarray(overlaps, dtype=OVERLAP_DTYPE)
        h = C.c_void_p()
        _check(lib().rvn_shard_piles(self._h, _p(lengths), lengths.shape[0], _p(overlaps), overlaps.shape[0], kmax,
                                     C.byref(h)))
        return Pass1(h, lengths.shape[0])

    # device-pointer variants (pointers are integers, e.g. torch.Tensor.data_ptr() of CUDA tensors)
    def shard_sketch_count(self, own_reads: Reads, index_minhash=False) -> int:
        n = C.c_uint64(0)
        _check(lib().rvn_shard_sketch(self._h, own_reads._h, int(index_minhash), C.byref(n)))
        return int(n.value)

    def shard_sketch_fetch_dev(self, d_values, d_origins):
        _check(lib().rvn_shard_sketch_fetch_dev(self._h, d_values, d_origins))

    def shard_index_build_dev(self, d_values, d_origins, n, all_query, n_flagged):
        _check(lib().rvn_shard_index_build_dev(self._h, d_values, d_origins, int(n), int(all_query), int(n_flagged)))

    def shard_key_histogram(self):
        hist = np.zeros(65536, dtype=np.uint64)
        over = np.zeros(1 << 20, dtype=np.uint32)
        n = C.c_uint32(0)
        _check(lib().rvn_shard_key_histogram(self._h, _p(hist), _p(over), over.shape[0], C.byref(n)))
        return hist.astype(np.int64), over[:n.value].astype(np.int64)

    def shard_join_count(self, n_reads_total, avoid_equal=True, avoid_symmetric=True, query_first=0, query_last=None) -> int:
        h = C.c_uint64(0)
        query_last = n_reads_total if query_last is None else query_last
        _check(lib().rvn_shard_join_range(self._h, n_reads_total, int(avoid_equal), int(avoid_symmetric), query_first,
                                          query_last, C.byref(h)))
        return int(h.value)

    # partition / regroup steps of the sharded pass on device pointers (shard.hip)
    def shard_split_minimizers_dev(self, d_val, d_org, n, world, d_val_out, d_org_out):
        counts = (C.c_uint64 * world)()
        L = lib()
        L.rvn_shard_split_minimizers_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                                     C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.rvn_shard_split_minimizers_dev(self._h, d_val, d_org, int(n), world, d_val_out, d_org_out, counts))
        return [int(x) for x in counts]

    def shard_count_flagged_dev(self, d_org, n) -> int:
        c = C.c_uint64(0)
        L = lib()
        L.rvn_shard_count_flagged_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        _check(L.rvn_shard_count_flagged_dev(self._h, d_org, int(n), C.byref(c)))
        return int(c.value)

    def shard_adjacent_diff_dev(self, d_seg, n, d_cnt):
        L = lib()
        L.rvn_shard_adjacent_diff_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        _check(L.rvn_shard_adjacent_diff_dev(self._h, d_seg, int(n), d_cnt))

    def shard_regroup_dev(self, d_cnt, d_grp, d_pos, n_src, n_reads, d_seg, d_grp_out, d_pos_out):
        world = len(d_cnt)
        arr = lambda xs: (C.c_void_p * world)(*[int(x) for x in xs])
        ns = (C.c_uint64 * world)(*[int(x) for x in n_src])
        L = lib()
        L.rvn_shard_regroup_dev.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.rvn_shard_regroup_dev(self._h, world, arr(d_cnt), arr(d_grp), arr(d_pos), ns, int(n_reads), d_seg,
                                       d_grp_out, d_pos_out))

    def shard_split_overlaps_dev(self, d_ovl, n, bounds, world, self_rank, d_out):
        b = np.ascontiguousarray(bounds, dtype=np.uint32)
        counts = (C.c_uint64 * (world + 1))()
        L = lib()
        L.rvn_shard_split_overlaps_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32,
                                                   C.c_void_p, C.c_void_p]
        _check(L.rvn_shard_split_overlaps_dev(self._h, d_ovl, int(n), _p(b), world, self_rank, d_out, counts))
        return [int(x) for x in counts]

    def shard_join_fetch_dev(self, d_grp, d_pos, d_seg):
        _check(lib().rvn_shard_join_fetch_dev(self._h, d_grp, d_pos, d_seg))

    def shard_chain_dev(self, own_reads: Reads, d_grp, d_pos, d_seg, n_matches) -> int:
        n = C.c_uint64(0)
        _check(lib().rvn_shard_chain_dev(self._h, own_reads._h, d_grp, d_pos, d_seg, int(n_matches), C.byref(n)))
        return int(n.value)

    def map_fetch_dev(self, d_overlaps, d_read_off):
        _check(lib().rvn_engine_map_fetch_dev(self._h, d_overlaps, d_read_off))

    def shard_piles_dev(self, lengths, d_overlaps, d_read_off, n, kmax=32) -> Pass1:
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        h = C.c_void_p()
        _check(lib().rvn_shard_piles_dev(self._h, _p(lengths), lengths.shape[0], d_overlaps, d_read_off, int(n), kmax,
                                         C.byref(h)))
        return Pass1(h, lengths.shape[0])

    def pile_add_layers(self, data: np.ndarray, pile_id: int, overlaps: np.ndarray):
        assert data.dtype == np.uint16 and overlaps.dtype == OVERLAP_DTYPE
        overlaps = np.ascontiguousarray(overlaps)
        _check(lib().rvn_pile_add_layers(self._h, _p(data), data.shape[0], pile_id, _p(overlaps),
                                         overlaps.shape[0]))

    # -- racon::Polisher::Polish, one round ------------------------------------------------------------
    def polish_round_range(self, targets: Reads, reads: Reads, window_first, window_last, quals=None, q=0.0, err=0.3,
                           w=500, trim=True, m=3, n=-5, g=-4):
        """One round restricted to global windows [window_first, window_last): returns (per-target partial consensus,
        per-target window counts in range, per-target polished counts, stats)."""
        nt = targets.n
        ooff = np.zeros(nt + 1, dtype=np.uint64)
        np.cumsum(2 * targets.rs.lengths.astype(np.uint64) + 1024, out=ooff[1:])
        out = np.zeros(int(ooff[-1]) + 1, dtype=np.uint8)
        out_len = np.zeros(nt, dtype=np.uint32)
        nw = np.zeros(nt, dtype=np.uint32)
        npol = np.zeros(nt, dtype=np.uint32)
        stats = np.zeros(16, dtype=np.uint64)
        qa = qo = None
        if quals is not None:
            qo = np.zeros(len(quals) + 1, dtype=np.uint64)
            np.cumsum([len(x) for x in quals], out=qo[1:])
            qa = np.concatenate([np.asarray(x, dtype=np.uint8) for x in quals])
        L = lib()
        L.rvn_polish_round_range.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                             C.c_double, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_uint64, C.c_uint64] + [C.c_void_p] * 7
        _check(L.rvn_polish_round_range(self._h, targets._h, reads._h, _p(qa), _p(qo), float(q), float(err), w,
                                        int(trim), m, n, g, int(window_first), int(window_last), _p(out), _p(ooff),
                                        _p(out_len), None, _p(nw), _p(npol), _p(stats)))
        cons = [out[int(ooff[i]): int(ooff[i]) + int(out_len[i])].copy() for i in range(nt)]
        st = {"n_windows": int(stats[3]), "n_layers": int(stats[2])}
        for i, k2 in enumerate(("poa_ms", "map_ms", "host_ms", "total_ms")):
            st[k2] = float(stats[6 + i: 7 + i].view(np.float64)[0])
        st["align_ms"] = float(stats[11:12].view(np.float64)[0])
        return cons, nw, npol, st

    def polish_map_best(self, targets: Reads, reads: Reads, read_first=0, read_last=None, err=0.3):
        """First step of a round for the reads [read_first, read_last): (best overlaps [n, 8] uint32 rows of rvn_overlap,
        best target index per read uint32 with 0xFFFFFFFF = unused, number of overlaps found)."""
        read_last = reads.n if read_last is None else int(read_last)
        n = read_last - int(read_first)
        best = np.zeros((max(n, 1), 8), dtype=np.uint32)
        bt = np.zeros(max(n, 1), dtype=np.uint32)
        n_ovl = C.c_uint64(0)
        L = lib()
        L.rvn_polish_map_best.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.rvn_polish_map_best(self._h, targets._h, reads._h, int(read_first), read_last, float(err), _p(best),
                                     _p(bt), C.byref(n_ovl)))
        return best[:n], bt[:n], int(n_ovl.value)

    def polish_set_best(self, best, best_target):
        """Hands the complete best-overlap table to the next polish_round / polish_round_range call (which then skips
        its own mapping)."""
        best = np.ascontiguousarray(best, dtype=np.uint32).reshape(-1, 8)
        bt = np.ascontiguousarray(best_target, dtype=np.uint32)
        assert best.shape[0] == bt.shape[0]
        L = lib()
        L.rvn_polish_set_best.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _check(L.rvn_polish_set_best(self._h, _p(best), _p(bt), int(bt.shape[0])))

    # -- second mapping pass and identity filter (construct.cc:316-491, :162-217) ------------------------------------
    def find_overlaps_and_repetitive_regions(self, reads: Reads, pile_begin, pile_end, pile_invalid, freq=0.001,
                                             kmer_len=None, identity=0.0, batch_bases=1 << 30):
        """raven::FindOverlapsAndRepetetiveRegions on the device.  pile_begin / pile_end in bases (Pile::begin() /
        end()), pile_invalid 0/1.  Returns dict(overlaps, contained[n], kmers (list of per-read uint8 cell arrays, empty
        for invalid reads))."""
        n = reads.n
        b = np.ascontiguousarray(pile_begin, dtype=np.uint32)
        en = np.ascontiguousarray(pile_end, dtype=np.uint32)
        inv = np.ascontiguousarray(pile_invalid, dtype=np.uint8)
        assert b.shape[0] == n and en.shape[0] == n and inv.shape[0] == n
        h = C.c_void_p()
        L = lib()
        _check(L.rvn_find_overlaps_and_repetitive_regions(self._h, reads._h, _p(b), _p(en), _p(inv), float(freq),
                                                          int(self.k if kmer_len is None else kmer_len), float(identity),
                                                          int(batch_bases), C.byref(h)))
        try:
            no, nk = int(L.rvn_pass2_num_overlaps(h)), int(L.rvn_pass2_kmer_cells(h))
            ovl = np.zeros(no, dtype=OVERLAP_DTYPE)
            contained = np.zeros(n, dtype=np.uint8)
            kmers = np.zeros(nk, dtype=np.uint8)
            koff = np.zeros(n + 1, dtype=np.uint64)
            _check(L.rvn_pass2_fetch(h, _p(ovl), _p(contained), _p(kmers), _p(koff)))
        finally:
            L.rvn_pass2_destroy(h)
        return dict(overlaps=ovl, contained=contained, kmers=[kmers[int(koff[i]):int(koff[i + 1])] for i in range(n)])

    def filter_overlaps_by_identity(self, reads: Reads, overlaps, offsets, pile_begin, pile_end, pile_invalid, identity):
        """Identity filter loop of ResolveContainedReads on per-pile lists: returns (overlaps, offsets) filtered."""
        o = np.ascontiguousarray(overlaps, dtype=OVERLAP_DTYPE).copy()
        off = np.ascontiguousarray(offsets, dtype=np.uint32).copy()
        _check(lib().rvn_filter_overlaps_by_identity(self._h, reads._h, _p(o), _p(off),
                                                     _p(np.ascontiguousarray(pile_begin, dtype=np.uint32)),
                                                     _p(np.ascontiguousarray(pile_end, dtype=np.uint32)),
                                                     _p(np.ascontiguousarray(pile_invalid, dtype=np.uint8)), float(identity)))
        return o[:int(off[-1])], off

    def release_scratch(self):
        _check(lib().rvn_engine_release_scratch(self._h))

    def poa_cells(self):
        """DP work of the banded POA kernel since the last reset_stats (rvn_poa_work)."""
        out = np.zeros(3, dtype=np.uint64)
        lib().rvn_poa_work(self._h, _p(out))
        return {"cells_full": int(out[0]), "cells_banded": int(out[1]), "calls": int(out[2])}

    def polish_layers(self):
        """Layer table of the last polishing round: uint32[n, 7] rows {window, read, first base in the oriented read,
        bases, begin, end, rc} in racon's order (rvn_polish_fetch_layers)."""
        n = C.c_uint64(0)
        _check(lib().rvn_polish_fetch_layers(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 7), dtype=np.uint32)
        _check(lib().rvn_polish_fetch_layers(self._h, _p(out), n.value, C.byref(n)))
        return out

    def polish_round(self, targets: Reads, reads: Reads, quals=None, q=0.0, err=0.3, w=500, trim=True, m=3, n=-5, g=-4):
        """quals: list of uint8 Phred+33 arrays (one per read) or None.  Returns (list of polished code arrays,
        ratio array, stats dict).  The engine must have k=15, w=5 (racon's mapping parameters)."""
        nt = targets.n
        ooff = np.zeros(nt + 1, dtype=np.uint64)
        np.cumsum(2 * targets.rs.lengths.astype(np.uint64) + 1024, out=ooff[1:])
        out = np.empty(int(ooff[-1]) + 1, dtype=np.uint8)  # (the library writes every byte it reports)
        out_len = np.zeros(nt, dtype=np.uint32)
        ratio = np.zeros(nt, dtype=np.float64)
        stats = np.zeros(16, dtype=np.uint64)
        qa = qo = None
        if quals is not None:
            qo = np.zeros(len(quals) + 1, dtype=np.uint64)
            np.cumsum([len(x) for x in quals], out=qo[1:])
            qa = np.concatenate([np.asarray(x, dtype=np.uint8) for x in quals])
        _check(lib().rvn_polish_round(self._h, targets._h, reads._h, _p(qa), _p(qo), float(q), float(err), w, int(trim),
                                      m, n, g, _p(out), _p(ooff), _p(out_len), _p(ratio), _p(stats)))
        # (views of this call's own buffer — never written again —, not copies: 100 MB at C4)
        cons = [out[int(ooff[i]): int(ooff[i]) + int(out_len[i])] for i in range(nt)]
        keys = ("n_overlaps", "n_reads_used", "n_layers", "n_windows", "n_polished_windows", "n_failed_windows")
        st = {k2: int(v) for k2, v in zip(keys, stats[:6])}
        for i, k2 in enumerate(("poa_ms", "map_ms", "host_ms", "total_ms")):
            st[k2] = float(stats[6 + i: 7 + i].view(np.float64)[0])
        st["n_dropped_layers"] = int(stats[10])
        st["align_ms"] = float(stats[11:12].view(np.float64)[0])
        st["n_aligned"], st["n_align_retries"] = int(stats[12]), int(stats[13])
        st["align_band_cells"], st["align_store_bytes"] = int(stats[14]), int(stats[15])
        return cons, ratio, st

    # -- raven::Pile::AddKmers, batched ---------------------------------------------------------------
    def pile_add_kmers_batch(self, reads: Reads, first, positions_per_read):
        """positions_per_read: list of uint32 arrays (the `filtered` output of Map per read, starting at read
        `first`).  Returns a list of uint8 arrays, Pile::kmers_ of each read ((len >> 4) + 1 entries)."""
        n = len(positions_per_read)
        poff = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum([len(x) for x in positions_per_read], out=poff[1:])
        pos = (np.concatenate([np.asarray(x, dtype=np.uint32) for x in positions_per_read])
               if n and poff[-1] else np.zeros(1, np.uint32))
        sizes = [(int(reads.rs.lengths[first + i]) >> 4) + 1 for i in range(n)]
        koff = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(sizes, out=koff[1:])
        out = np.zeros(int(koff[-1]) + 1, dtype=np.uint8)
        _check(lib().rvn_pile_add_kmers_batch(self._h, reads._h, first, n, _p(pos), _p(poff), _p(out), _p(koff)))
        return [out[int(koff[i]): int(koff[i + 1])].copy() for i in range(n)]

    # -- edlibAlign(default config).editDistance, batched -----------------------------------------
    def edit_distance_batch(self, reads: Reads, pairs: np.ndarray):
        """pairs: array of ED_PAIR_DTYPE; returns (uint32 distances, device ms, DP cells)."""
        pairs = np.ascontiguousarray(pairs, dtype=ED_PAIR_DTYPE)
        out = np.zeros(pairs.shape[0], dtype=np.uint32)
        ms, cells = C.c_double(0), C.c_uint64(0)
        _check(lib().rvn_edit_distance_batch(self._h, reads._h, _p(pairs), pairs.shape[0], _p(out), C.byref(ms),
                                             C.byref(cells)))
        return out, ms.value, cells.value

    # -- racon Window::GenerateConsensus, batched --------------------------------------------------
    def poa_consensus_batch(self, windows, m=3, n=-5, g=-4, trim=True):
        """windows: list of dicts {layers: [uint8 code arrays, layer 0 = backbone], begins, ends, quals (list of
        uint8 Phred+33 arrays or None entries) or None}.  Returns (list of consensus code arrays, status array, ms)."""
        a = _pack_poa_windows(windows)
        ms = C.c_double(0)
        _check(lib().rvn_poa_consensus_batch(
            self._h, _p(a["codes"]), _p(a["quals"]), _p(a["loff"]), _p(a["begins"]), _p(a["ends"]), _p(a["hasq"]),
            _p(a["woff"]), a["nw"], m, n, g, int(trim), _p(a["out"]), _p(a["ooff"]), _p(a["out_len"]), _p(a["status"]),
            C.byref(ms)))
        return _unpack_poa_consensus(a), a["status"], ms.value

    def poa_phase_cycles(self):
        c = np.zeros(6, dtype=np.uint64)
        lib().rvn_poa_phase_cycles(self._h, _p(c))
        return dict(zip(("subgraph", "dp", "traceback", "add_alignment", "order", "consensus"), (int(x) for x in c)))

    def polish_set_chunk_windows(self, windows):
        """Windows per POA chunk of a polishing round (0 = one batch); returns the previous value."""
        return int(lib().rvn_polish_set_chunk_windows(self._h, int(windows)))

    def set_option(self, name, value):
        """rvn_engine_set_option: a tuning option (include/raven_hip.h lists them; -1 = built-in default, and so is 0 for
        every option but poa_rows_min_windows, whose 0 means "every batch" and selects another first kernel: consensus is
        reproducible per value of that option, not across values); returns the previous value.  Unknown names raise."""
        prev = C.c_int64(0)
        L = lib()
        L.rvn_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p]
        _check(L.rvn_engine_set_option(self._h, name.encode(), int(value), C.byref(prev)))
        return int(prev.value)

    def poa_set_mode(self, mode):
        """0 band 32 (rows on lanes, poa4.hip) -> 64 -> 128 -> 256 -> full matrix (default), 1 full matrix only,
        2 / 3 / 4 band 64 / 128 / 256 only, 9 poa4.hip only."""
        return int(lib().rvn_poa_set_mode(self._h, int(mode)))

    def poa_fallback_windows(self):
        return int(lib().rvn_poa_fallback_windows(self._h))

    def poa_wide_windows(self):
        return int(lib().rvn_poa_wide_windows(self._h))

    def poa_narrow_windows(self):
        return int(lib().rvn_poa_narrow_windows(self._h))

    # -- introspection ---------------------------------------------------------------------
    def sketch(self, reads: Reads, first=0, last=None, minhash=False):
        last = reads.n if last is None else last
        cnt = C.c_uint64(0)
        _check(lib().rvn_engine_sketch(self._h, reads._h, first, last, int(minhash), C.byref(cnt)))
        v = np.zeros(cnt.value, dtype=np.uint64)
        o = np.zeros(cnt.value, dtype=np.uint64)
        off = np.zeros(last - first + 1, dtype=np.uint32)
        _check(lib().rvn_engine_sketch_fetch(self._h, _p(v), _p(o), _p(off)))
        return v, o, off

    def index_content(self):
        m, u = C.c_uint64(0), C.c_uint64(0)
        _check(lib().rvn_engine_index_size(self._h, C.byref(m), C.byref(u)))
        v = np.zeros(m.value, dtype=np.uint64)
        o = np.zeros(m.value, dtype=np.uint64)
        _check(lib().rvn_engine_index_fetch(self._h, _p(v), _p(o)))
        return v, o, int(u.value)

    def counters(self):
        c = np.zeros(8, dtype=np.uint64)
        _check(lib().rvn_engine_counters(self._h, _p(c)))
        return dict(zip(("index_bases", "index_minimizers", "index_keys", "query_bases", "query_minimizers",
                         "matches", "overlaps", "intervals"), (int(x) for x in c)))

    def stage_ms(self):
        L = lib()
        n = L.rvn_engine_num_stages()
        ms = np.zeros(n, dtype=np.float64)
        la = np.zeros(n, dtype=np.uint64)
        _check(L.rvn_engine_stage_ms(self._h, _p(ms), _p(la), n))
        return {L.rvn_engine_stage_name(i).decode(): (float(ms[i]), int(la[i])) for i in range(n)}

    def set_kernel_timing(self, enabled: bool):
        lib().rvn_engine_set_kernel_timing(self._h, int(enabled))

    def kernel_ms(self):
        """{site: (total device ms, launches)} since the last reset_stats()."""
        L = lib()
        n = L.rvn_engine_num_kernel_sites()
        ms = np.zeros(n, dtype=np.float64)
        la = np.zeros(n, dtype=np.uint64)
        _check(L.rvn_engine_kernel_ms(self._h, _p(ms), _p(la), n))
        return {L.rvn_engine_kernel_site_name(i).decode(): (float(ms[i]), int(la[i])) for i in range(n)}

    def reset_stats(self):
        lib().rvn_engine_reset_stats(self._h)

    def set_timing(self, enabled: bool):
        lib().rvn_engine_set_timing(self._h, int(enabled))


def test_find_chimeric_regions(data):
    """slopes.h on the host: Pile::FindChimericRegions of one coverage array -> (k, 2) uint32 (begin, end) cells."""
    d = np.ascontiguousarray(data, dtype=np.uint16)
    out = np.zeros(max(2, d.shape[0]), dtype=np.uint32)
    n = test_lib().rvn_test_find_chimeric_regions(_p(d), d.shape[0], _p(out), out.shape[0] // 2)
    if n < 0:
        raise ValueError("rvn_test_find_chimeric_regions: %d" % n)
    return out[:2 * n].reshape(-1, 2).copy()


def overlap_update_and_type(overlaps, pile_begin, pile_end, pile_invalid):
    """rvn_overlap_update_and_type: raven::OverlapUpdate + GetOverlapType on host arrays (overlap_utils.cc:14-121):
    (updated overlaps, ok, type)."""
    o = np.ascontiguousarray(overlaps, dtype=OVERLAP_DTYPE).copy()
    b = np.ascontiguousarray(pile_begin, dtype=np.uint32)
    ok = np.zeros(o.shape[0], dtype=np.uint8)
    ty = np.zeros(o.shape[0], dtype=np.uint32)
    _check(lib().rvn_overlap_update_and_type(_p(o), o.shape[0], _p(b), _p(np.ascontiguousarray(pile_end, dtype=np.uint32)),
                                             _p(np.ascontiguousarray(pile_invalid, dtype=np.uint8)), b.shape[0], _p(ok), _p(ty)))
    return o, ok, ty


def test_overlap_update_and_type(overlaps, pile_begin, pile_end, pile_invalid):
    """overlap_rules.h on the host (the __host__ __device__ code of the kernels): (updated overlaps, ok, type)."""
    o = np.ascontiguousarray(overlaps, dtype=OVERLAP_DTYPE).copy()
    b = np.ascontiguousarray(pile_begin, dtype=np.uint32)
    ok = np.zeros(o.shape[0], dtype=np.uint8)
    ty = np.zeros(o.shape[0], dtype=np.uint32)
    rc = test_lib().rvn_test_overlap_update_and_type(_p(o), o.shape[0], _p(b), _p(np.ascontiguousarray(pile_end, dtype=np.uint32)),
                                                _p(np.ascontiguousarray(pile_invalid, dtype=np.uint8)), b.shape[0], _p(ok),
                                                _p(ty))
    if rc != 0:
        raise ValueError("rvn_test_overlap_update_and_type")
    return o, ok, ty


def test_parse_file(path, fastq, threads=0, force_streaming=False, slab_bytes=0):
    """TEST INFRASTRUCTURE: the host half of rvn_reads_load (member cut, inflate pool, record scanner) without a device.
    Returns (names, list of base strings, list of quality strings or None, info dict)."""
    T = test_lib()
    b, q, l, nm = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    n = C.c_uint32(0)
    info = np.zeros(8, dtype=np.uint32)
    rc = T.rvn_test_parse_file(os.fsencode(path), int(fastq), threads, int(force_streaming), slab_bytes, C.byref(b),
                               C.byref(q), C.byref(l), C.byref(n), C.byref(nm), _p(info))
    if rc != RVN_OK:
        raise (ValueError if rc == RVN_EINVAL else RavenHipError)(T.rvn_last_error().decode(errors="replace"))
    try:
        lens = np.ctypeslib.as_array(C.cast(l, C.POINTER(C.c_uint32)), shape=(max(n.value, 1),))[:n.value].copy()
        total = int(lens.sum())
        bases = C.string_at(b, total)
        quals = C.string_at(q, total) if fastq else None
        names = C.string_at(nm).decode().split("\n")[:-1] if n.value else []
    finally:
        for x in (b, q, l, nm):
            T.rvn_free(x)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    seqs = [bases[off[i]:off[i + 1]] for i in range(n.value)]
    qs = [quals[off[i]:off[i + 1]] for i in range(n.value)] if fastq else None
    return names, seqs, qs, dict(gzip=int(info[0]), streaming=int(info[1]), members=int(info[2]), threads=int(info[3]),
                                 restarted=int(info[4]), loop_s=info[5] / 1e6, scan_s=info[6] / 1e6, fast=int(info[7]))


NW_REC_DTYPE = np.dtype([("first_t", "<u4"), ("first_q", "<u4"), ("last_t", "<u4"), ("last_q", "<u4"),
                         ("grid", "<u2", (8,))])


def test_nw_breakpoints(t_words, t_len, r_words, r_len, t_begin, n, q_begin, m, rc, w, k=64, force_r=0, group_lanes=0):
    """nwpath.h stepped on the CPU (no GPU needed): the forward sweep's lane code for 64 emulated lanes + the
    traceback (group_lanes = 4 / 16 / 64: the walk by a group of lanes per alignment, nwtrace.h).  Returns (records per
    window, exact distance, (k, lanes, R[, batches of the group walk]), status)."""
    t_words = np.ascontiguousarray(t_words, dtype=np.uint64)
    r_words = np.ascontiguousarray(r_words, dtype=np.uint64)
    n_win = (t_begin + n - 1) // w - t_begin // w + 1
    recs = np.zeros(n_win, dtype=NW_REC_DTYPE)
    dist = np.zeros(1, dtype=np.uint32)
    band = np.zeros(4 if group_lanes else 3, dtype=np.uint32)
    rc_ = test_lib().rvn_test_nw_breakpoints(_p(t_words), t_len, _p(r_words), r_len, t_begin, n, q_begin, m,
                                        (1 if rc else 0) | (int(group_lanes) << 8), w, k,
                                        force_r, _p(recs), _p(dist), _p(band))
    if rc_ < 0:
        raise ValueError("rvn_test_nw_breakpoints: %d" % rc_)
    return recs, int(dist[0]), tuple(int(x) for x in band), rc_
