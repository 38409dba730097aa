"""CPU tests of the __host__ __device__ building blocks of the HIP library through its rvn_test_* hooks
(no GPU needed: the hooks run the very same inline functions on the host)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from raven_amd import hip, seqio
from tests import refimpl


def test_hash32_equals_hash64():
    L = hip.test_lib()
    rng = np.random.default_rng(0)
    for k in (1, 5, 11, 15):
        mask = (1 << (2 * k)) - 1
        keys = rng.integers(0, mask + 1, size=500, dtype=np.uint64)
        want = refimpl._hash(keys, mask)
        for key, w in zip(keys.tolist(), want.tolist()):
            assert L.rvn_test_hash(key, k, 0) == w
            assert L.rvn_test_hash(key, k, 1) == w
    for k in (16, 19, 31):
        mask = (1 << (2 * k)) - 1
        keys = rng.integers(0, mask + 1, size=500, dtype=np.uint64)
        want = refimpl._hash(keys, mask)
        for key, w in zip(keys.tolist(), want.tolist()):
            assert L.rvn_test_hash(key, k, 0) == w


@pytest.mark.parametrize("k", [1, 7, 15, 16, 21, 31])
def test_canonical_kmer_extraction(k):
    """Direct extraction from the packed stream == ram's rolling forward/reverse registers."""
    L = hip.test_lib()
    rng = np.random.default_rng(k)
    codes = rng.integers(0, 4, size=400, dtype=np.uint8)
    codes[100:100 + 2 * k] = np.tile(np.array([0, 3], np.uint8), k)  # (AT)n -> palindromes when k is even
    words = np.concatenate([seqio.pack_codes(codes), np.zeros(2, np.uint64)])
    mask = (1 << (2 * k)) - 1
    P = codes.shape[0] - k + 1
    c = codes.astype(np.uint64)
    fwd = np.zeros(P, np.uint64)
    rev = np.zeros(P, np.uint64)
    for j in range(k):
        fwd |= c[j:j + P] << np.uint64(2 * (k - 1 - j))
        rev |= (np.uint64(3) - c[j:j + P]) << np.uint64(2 * j)
    want_h = refimpl._hash(np.minimum(fwd, rev), mask)
    for use32 in ((0, 1) if 2 * k < 32 else (0,)):
        for p in range(P):
            v, s = C.c_uint64(0), C.c_uint32(0)
            ok = L.rvn_test_canonical(words.ctypes.data_as(C.c_void_p), p, k, use32, C.byref(v), C.byref(s))
            if fwd[p] == rev[p]:
                assert ok == 0
            else:
                assert ok == 1 and v.value == int(want_h[p]) and s.value == int(fwd[p] > rev[p]), (p, use32)


def _sort_via_oracle(lens):
    import ctypes
    n = lens.shape[0]
    ovl = np.zeros(n, oracle.OVERLAP_DTYPE)
    ovl["lhs_end"] = lens
    ovl["score"] = np.arange(n)
    oracle.lib().orc_truncate(ovl.ctypes.data_as(ctypes.c_void_p), n, 1)  # sorts in place whenever n >= 1
    return ovl["score"].copy()


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 31, 32, 33, 64, 100, 257, 1000, 5000])
def test_device_introsort_equals_std_sort(n):
    """rvn::std_sort must reproduce libstdc++'s unstable std::sort permutation exactly (ties included)."""
    L = hip.test_lib()
    rng = np.random.default_rng(n)
    for spread in (1, 3, 20, 10 ** 6):
        for pattern in ("random", "sorted", "reversed", "organ"):
            lens = rng.integers(0, spread, size=n).astype(np.uint32)
            if pattern == "sorted":
                lens.sort()
            elif pattern == "reversed":
                lens[::-1].sort()
            elif pattern == "organ":
                lens = np.concatenate([np.sort(lens[: n // 2]), np.sort(lens[n // 2:])[::-1]]).astype(np.uint32)
            keys = (lens.astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
            mine = keys.copy()
            L.rvn_test_std_sort_lendesc(mine.ctypes.data_as(C.c_void_p), n)
            want = _sort_via_oracle(lens)
            assert np.array_equal((mine & np.uint64(0xFFFFFFFF)).astype(np.uint32), want), (n, spread, pattern)


@pytest.mark.parametrize("n", [100, 1000, 20000])
def test_device_introsort_depth_limit_path(n):
    """Adversarial input (McIlroy) forces std::sort's depth-limit heapsort fallback; permutations must match."""
    L = hip.test_lib()
    vals = oracle.antiqsort(n)
    lens = (np.uint32(n + 5) - vals).astype(np.uint32)  # descending comparator sees the ascending killer
    keys = (lens.astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    L.rvn_test_std_sort_lendesc(keys.ctypes.data_as(C.c_void_p), n)
    want = _sort_via_oracle(lens)
    assert np.array_equal((keys & np.uint64(0xFFFFFFFF)).astype(np.uint32), want)
    # and with heavy ties on top of the killer order
    lens2 = (lens // 7).astype(np.uint32)
    keys = (lens2.astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    L.rvn_test_std_sort_lendesc(keys.ctypes.data_as(C.c_void_p), n)
    assert np.array_equal((keys & np.uint64(0xFFFFFFFF)).astype(np.uint32), _sort_via_oracle(lens2))


def test_device_heapsort_is_a_sort():
    L = hip.test_lib()
    rng = np.random.default_rng(1)
    for n in (0, 1, 2, 3, 17, 100, 1001):
        lens = rng.integers(0, 50, size=n).astype(np.uint64)
        keys = (lens << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        L.rvn_test_heap_sort_lendesc(keys.ctypes.data_as(C.c_void_p), n)
        got = (keys >> np.uint64(32)).astype(np.int64)
        assert np.all(np.diff(got) <= 0)
        assert sorted((keys & np.uint64(0xFFFFFFFF)).tolist()) == list(range(n))


@pytest.mark.parametrize("k", [5, 15, 16, 19, 31])
def test_low_complexity_filter_matches_pile_cc(k):
    """rvn::lc_kmer_passes (used by the AddKmers kernel) vs the literal std::string restatement of
    RavenLib/src/pile.cc:73-117 in the oracle, on random, homopolymer-rich and short-period k-mers."""
    L = hip.test_lib()
    rng = np.random.default_rng(k)
    reads = []
    for period in (1, 2, 3, 4, 5, 7):
        unit = rng.integers(0, 4, size=period, dtype=np.uint8)
        rep = np.tile(unit, 400 // period + 1)[:400]
        noise = rng.random(400) < 0.08
        rep[noise] = rng.integers(0, 4, size=int(noise.sum()), dtype=np.uint8)
        reads.append(rep)
    reads.append(rng.integers(0, 4, size=600, dtype=np.uint8))
    runs = np.repeat(rng.integers(0, 4, size=200, dtype=np.uint8), rng.integers(1, 6, size=200))
    reads.append(runs[:600].astype(np.uint8))
    rs = seqio.pack_reads(reads)
    n_pass = n_fail = 0
    for i in range(rs.n):
        codes = rs.codes(i)
        for p in range(0, codes.shape[0] - k + 1, 3):
            want = oracle.pile_add_kmers(rs, i, np.array([p], np.uint32), k)
            kc = np.ascontiguousarray(codes[p:p + k])
            got = L.rvn_test_low_complexity(kc.ctypes.data_as(C.c_void_p), k)
            assert got == int(want[p >> 4]), (i, p, kc.tolist())
            n_pass += got
            n_fail += 1 - got
    assert n_pass > 20 and n_fail > 20
