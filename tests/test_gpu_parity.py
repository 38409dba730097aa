"""GPU parity tests (pytest -m gpu): every stage of the HIP engine, called through the C ABI
(raven_amd/hip.py -> libraven_hip.so), bit-exact against the CPU oracle on the same seeded inputs, plus
size-independent properties at the full BASELINE configs[1] size."""
import numpy as np
import pytest

from oracle import oracle
from raven_amd import hip, seqio, synth
from tests import parity_util as pu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if hip.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests must run on the MI355X box")
    return True


def _mk(k=15, w=5, **kw):
    return hip.Engine(k, w, **kw), oracle.Engine(k, w, **kw)


@pytest.mark.parametrize("minhash", [False, True])
def test_sketch_lambda(gpu, lambda_reads, minhash):
    he, oe = _mk()
    rd = he.upload(lambda_reads)
    assert pu.compare_sketch(he, oe, rd, lambda_reads, 0, lambda_reads.n, minhash) == []


@pytest.mark.parametrize("k,w", [(15, 5), (5, 3), (11, 33), (16, 4), (19, 7), (31, 10), (15, 1), (13, 256)])
def test_sketch_edge_shapes(gpu, k, w):
    """read < k, == k, just below / at k+w-1, tile boundaries (1024 positions), homopolymers, palindromic repeats."""
    rng = np.random.default_rng(k * 1000 + w)
    lens = [0, 1, k - 1, k, k + w - 2, k + w - 1, k + w, 1023 + k, 1024 + k - 1, 1024 + k, 1025 + k, 2048 + k + w, 5000,
            61234]
    reads = [rng.integers(0, 4, size=n, dtype=np.uint8) for n in lens]
    reads += [np.zeros(3000, np.uint8), np.tile(np.array([0, 3], np.uint8), 1500),
              np.tile(np.array([0, 1, 2, 3], np.uint8), 700), np.full(1500, 2, np.uint8)]
    rs = seqio.pack_reads(reads)
    he, oe = _mk(k, w)
    rd = he.upload(rs)
    for mh in (False, True):
        assert pu.compare_sketch(he, oe, rd, rs, 0, rs.n, mh) == []


@pytest.mark.parametrize("minhash", [False, True])
def test_index_filter_map_lambda(gpu, lambda_reads, minhash):
    rs = lambda_reads
    he, oe = _mk()
    rd = he.upload(rs)
    he.minimize(rd, 0, rs.n, minhash)
    oe.minimize(rs, 0, rs.n, minhash)
    assert pu.compare_index(he, oe, rs, 0, rs.n, minhash) == []
    for f in (0.001, 0.01, 0.2, 0.0, 1.0):
        he.filter(f)
        oe.filter(f)
        assert he.occurrence == oe.occurrence, f
    he.filter(0.001)
    oe.filter(0.001)
    errs, n = pu.compare_map(he, oe, rd, rs, 0, rs.n, True)
    assert errs == [] and n > 1000
    errs, _ = pu.compare_map(he, oe, rd, rs, 0, 64, False, want_filtered=True)
    assert errs == []
    # avoid_equal / avoid_symmetric off (racon-style mapping flags)
    errs, _ = pu.compare_map(he, oe, rd, rs, 0, 40, True, avoid_equal=False, avoid_symmetric=False)
    assert errs == []
    errs, _ = pu.compare_map(he, oe, rd, rs, 10, 50, False, avoid_equal=True, avoid_symmetric=False)
    assert errs == []


def test_filter_rejects_bad_frequency(gpu, lambda_reads):
    he, _ = _mk()
    with pytest.raises(ValueError):
        he.filter(1.5)
    with pytest.raises(ValueError):
        he.filter(-0.01)


def test_index_subrange_and_heavy_filter(gpu, synth_small):
    """Index over a sub-range of reads, queries over another; f large enough that the filter bites."""
    _, rs, _ = synth_small
    he, oe = _mk()
    rd = he.upload(rs)
    he.minimize(rd, 100, 300, False)
    oe.minimize(rs, 100, 300, False)
    assert pu.compare_index(he, oe, rs, 100, 300, False) == []
    he.filter(0.05)
    oe.filter(0.05)
    assert he.occurrence == oe.occurrence
    errs, _ = pu.compare_map(he, oe, rd, rs, 0, 160, False, want_filtered=True)
    assert errs == []


@pytest.mark.parametrize("kw", [
    dict(use_minhash=False), dict(use_minhash=True), dict(use_minhash=False, kmax=8),
    dict(use_minhash=False, kmax=1), dict(use_minhash=False, freq=0.0)])
def test_pass1_lambda(gpu, lambda_reads, kw):
    he, oe = _mk()
    rd = he.upload(lambda_reads)
    errs, _ = pu.compare_pass1(he, oe, rd, lambda_reads, **kw)
    assert errs == []


def test_pass1_and_map_with_every_value_addressed_directly(gpu, synth_small, lambda_reads):
    """Large indexes address all 4^k possible values directly (index.hip: one cache line per probe instead of bucket table ->
    search -> run table); engine option index_direct_min_keys = 1 takes that path on small inputs: per-read Map output and the
    whole pass — several index batches and query flushes, so the probe path runs, not the self-join — equal the oracle's."""
    _, rs, _ = synth_small
    for reads, kw in ((rs, dict(index_batch_bases=rs.total_bases // 2, flush_bases=rs.total_bases // 5 + 1)),
                      (rs, dict(flush_bases=rs.total_bases // 3 + 1, use_minhash=True)),
                      (lambda_reads, dict(flush_bases=lambda_reads.total_bases // 4 + 1, kmax=8))):
        he, oe = _mk()
        he.set_option("index_direct_min_keys", 1)
        rd = he.upload(reads)
        errs, ref = pu.compare_pass1(he, oe, rd, reads, **kw)
        assert errs == [], kw
        assert ref["counters"]["matches"] > 0
    he, oe = _mk()
    he.set_option("index_direct_min_keys", 1)
    rd = he.upload(rs)
    he.minimize(rd, 100, 300, False)  # (index over a sub-range of the reads, queries over another, a filter that bites)
    oe.minimize(rs, 100, 300, False)
    he.filter(0.05)
    oe.filter(0.05)
    assert he.occurrence == oe.occurrence
    errs, n_ovl = pu.compare_map(he, oe, rd, rs, 0, 160, False, want_filtered=True)
    assert errs == [] and n_ovl > 0
    errs, _ = pu.compare_map(he, oe, rd, rs, 120, 260, True, avoid_equal=False, avoid_symmetric=False)
    assert errs == []


def test_pass1_multibatch(gpu, synth_small):
    """several index batches x several query flushes (construct.cc:32-37, :66-70 with small constants)"""
    _, rs, _ = synth_small
    for ib, fb in ((rs.total_bases // 3 + 1, rs.total_bases // 7 + 1), (rs.total_bases // 2, 1 << 40),
                   (1 << 40, rs.total_bases // 4)):
        he, oe = _mk()
        rd = he.upload(rs)
        errs, _ = pu.compare_pass1(he, oe, rd, rs, index_batch_bases=ib, flush_bases=fb)
        assert errs == [], (ib, fb)


def test_pass1_k19_hifi_like(gpu):
    g = synth.make_genome(150_000, seed=21)
    rs, _ = synth.make_reads(g, 15, 9000, length_model="lognormal", seed=22, sub=0.001, ins=0.002, dele=0.002)
    he, oe = _mk(19, 7)
    rd = he.upload(rs)
    errs, ref = pu.compare_pass1(he, oe, rd, rs)
    assert errs == []
    assert ref["counters"]["matches"] > 10 * ref["counters"]["overlaps"]


def test_pass1_long_intervals(gpu):
    """30 kb low-error reads: intervals with > 1024 matches take the chain kernel's global-scratch path."""
    g = synth.make_genome(120_000, seed=51)
    rs, _ = synth.make_reads(g, 10, 30000, seed=52, sub=0.001, ins=0.0005, dele=0.0005)
    he, oe = _mk()
    rd = he.upload(rs)
    errs, ref = pu.compare_pass1(he, oe, rd, rs)
    assert errs == []
    c = ref["counters"]
    assert c["matches"] / max(1, c["overlaps"]) > 400
    # and the un-minhashed query side (5x more matches per interval)
    he.minimize(rd, 0, rs.n, False)
    oe.minimize(rs, 0, rs.n, False)
    he.filter(0.001)
    oe.filter(0.001)
    errs, _ = pu.compare_map(he, oe, rd, rs, 0, rs.n, False)
    assert errs == []


def test_pass1_ragged_and_empty(gpu):
    rng = np.random.default_rng(5)
    g = synth.make_genome(60_000, seed=31)
    rs0, _ = synth.make_reads(g, 10, 3000, length_model="lognormal", seed=32, min_len=200)
    reads = [rs0.codes(i) for i in range(rs0.n)]
    reads.insert(3, np.zeros(0, np.uint8))          # empty read
    reads.insert(7, rng.integers(0, 4, size=10, dtype=np.uint8))  # shorter than k
    reads.insert(9, rng.integers(0, 4, size=18, dtype=np.uint8))  # k <= len < k+w-1
    reads.append(np.zeros(2500, np.uint8))          # homopolymer
    reads.append(reads[0].copy())                   # exact duplicate of read 0
    rs = seqio.pack_reads(reads)
    he, oe = _mk()
    rd = he.upload(rs)
    errs, _ = pu.compare_pass1(he, oe, rd, rs)
    assert errs == []
    # zero reads / a single read
    for sub in ([], reads[:1]):
        rs1 = seqio.pack_reads(sub)
        he, oe = _mk()
        rd = he.upload(rs1)
        p = he.find_overlaps_and_create_piles(rd)
        assert p.overlaps()[0].shape[0] == 0
        p.close()


def test_add_layers_c_abi(gpu):
    rng = np.random.default_rng(9)
    he, _ = _mk()
    for trial in range(16):
        # trials 12..15: deep piles (several thousand overlaps in ONE AddLayers call, saturating), piles longer than
        # the kernel's LDS tile (8192 cells), and overlaps shorter than 32 bases whose end event precedes their begin
        # event (the reference's uint32 coverage wraps around, pile.cc:60)
        L = int(rng.integers(200, 60000)) if trial < 12 else int(rng.integers(150_000, 400_000))
        cells = L >> 4
        n = int(rng.integers(1, 3000 if trial == 0 else 200)) if trial < 12 else int(rng.integers(3000, 9000))
        ovl = np.zeros(n, hip.OVERLAP_DTYPE)
        b = rng.integers(0, max(1, L - 120), size=n)
        en = np.minimum(L, b + rng.integers(100, L, size=n))
        if trial >= 14:
            tiny = (rng.random(n) < 0.02) & (b >= 16)  # (end >> 4) - 1 must not underflow (UB in the reference)
            en = np.where(tiny, np.minimum(L, b + rng.integers(1, 31, size=n)), en)
        side = rng.integers(0, 3, size=n)
        ovl["lhs_id"] = np.where(side == 0, 42, 7)
        ovl["rhs_id"] = np.where(side == 1, 42, 8)
        ovl["lhs_begin"], ovl["lhs_end"] = np.where(side == 0, b, 5), np.where(side == 0, en, 500)
        ovl["rhs_begin"], ovl["rhs_end"] = np.where(side == 1, b, 9), np.where(side == 1, en, 900)
        data = rng.integers(0, 50, size=cells).astype(np.uint16)
        if trial % 4 == 0:
            data[:] = 65500 if trial < 12 else 62000  # saturation at 65535
        want = data.copy()
        oracle.pile_add_layers(want, 42, ovl.astype(oracle.OVERLAP_DTYPE))
        got = data.copy()
        he.pile_add_layers(got, 42, ovl)
        assert np.array_equal(got, want), trial


def test_full_size_properties(gpu):
    """BASELINE configs[1] size (5 Mb, 30x, 10 kb): size-independent properties instead of the (slow) oracle."""
    g = synth.make_genome(5_000_000)
    rs, truth = synth.make_reads(g, 30, 10000)
    he, _ = _mk()
    rd = he.upload(rs)
    p = he.find_overlaps_and_create_piles(rd)
    ovl, off = p.overlaps()
    data, poff = p.piles()
    c = he.counters()
    p.close()
    # idempotence: a second pass gives byte-identical results
    p2 = he.find_overlaps_and_create_piles(rd)
    ovl2, off2 = p2.overlaps()
    data2, _ = p2.piles()
    p2.close()
    assert np.array_equal(ovl, ovl2) and np.array_equal(off, off2) and np.array_equal(data, data2)
    # structure: CSR consistent, <= kmax per pile, lists sorted by length once truncated, ids consistent
    assert off[0] == 0 and off[-1] == ovl.shape[0] and np.all(np.diff(off.astype(np.int64)) <= 32)
    pile_of = np.repeat(np.arange(rs.n, dtype=np.uint32), np.diff(off.astype(np.int64)))
    assert np.array_equal(ovl["lhs_id"], pile_of) and np.all(ovl["rhs_id"] != ovl["lhs_id"])
    assert np.all(ovl["lhs_end"] <= rs.lengths[ovl["lhs_id"]]) and np.all(ovl["rhs_end"] <= rs.lengths[ovl["rhs_id"]])
    assert np.all(ovl["score"] >= 100)
    full = np.nonzero(np.diff(off.astype(np.int64)) == 32)[0]
    assert full.size > rs.n // 2
    lens = np.maximum(ovl["lhs_end"] - ovl["lhs_begin"], ovl["rhs_end"] - ovl["rhs_begin"]).astype(np.int64)
    for pidx in full[:2000]:
        assert np.all(np.diff(lens[off[pidx]: off[pidx + 1]]) <= 0)
    # symmetry checksum before truncation is impossible after the cut, so check against ground truth instead:
    s, e_ = truth["start"], truth["start"] + truth["src_len"]
    inter = np.minimum(e_[ovl["lhs_id"]], e_[ovl["rhs_id"]]) - np.maximum(s[ovl["lhs_id"]], s[ovl["rhs_id"]])
    assert (inter > 0).mean() > 0.999
    same = truth["strand"][ovl["lhs_id"]] == truth["strand"][ovl["rhs_id"]]
    assert (same == (ovl["strand"] == 1))[inter > 500].mean() > 0.999
    # pile coverage: total coverage cells == sum over ALL Map overlaps (both sides) of their cell spans is not
    # recoverable after truncation either; check saturation-free and plausibility (30x, both directions)
    assert data.max() < 65535 and 15 < data.mean() < 40
    assert c["index_bases"] == rs.total_bases and c["query_bases"] == rs.total_bases


def test_second_pass_pieces_filtered_and_add_kmers(gpu):
    """Device pieces of FindOverlapsAndRepetetiveRegions (construct.cc:363-382): full (non-minhash) index,
    Map(..., minhash=false, &filtered) and Pile::AddKmers on the filtered positions, on a genome with repeats."""
    g = synth.make_genome(60_000, seed=61)
    rep = g[1000:3000].copy()
    for off in (10_000, 25_000, 40_000, 52_000):  # 2 kb repeat, 5 copies -> high-occurrence minimizers
        g[off:off + 2000] = rep
    g[30_000:30_400] = 0                            # a homopolymer island (low complexity)
    rs, _ = synth.make_reads(g, 15, 5000, seed=62, sub=0.01, ins=0.005, dele=0.005)
    he, oe = _mk()
    rd = he.upload(rs)
    he.minimize(rd, 0, rs.n, False)
    oe.minimize(rs, 0, rs.n, False)
    he.filter(0.02)
    oe.filter(0.02)
    assert he.occurrence == oe.occurrence
    res = he.map_batch(rd, 0, rs.n, True, True, False, want_filtered=True)
    per_read = [res["filtered"][res["filtered_offsets"][i]: res["filtered_offsets"][i + 1]] for i in range(rs.n)]
    assert sum(len(x) for x in per_read) > 100
    got = he.pile_add_kmers_batch(rd, 0, per_read)
    marked = 0
    for i in range(rs.n):
        ref = oe.map(rs, i, True, True, False)
        assert np.array_equal(per_read[i], ref["filtered"])
        want = oracle.pile_add_kmers(rs, i, ref["filtered"], 15)
        assert np.array_equal(got[i], want), i
        marked += int(want.sum())
    assert marked > 10
    with pytest.raises(ValueError):
        he.pile_add_kmers_batch(rd, 0, [np.array([int(rs.lengths[0])], np.uint32)])


def test_pile_trim_and_median_on_device_matches_oracle():
    """Pile::FindValidRegion(4) + FindMedian on the coverage arrays in HBM
    (rvn_pass1_trim_and_annotate) vs the restatement of pile.cc:122-174, on the piles of a real pass plus
    low-coverage and long reads (invalid piles, runs without a terminator, multi-chunk piles)."""
    g = synth.make_genome(150_000, seed=77)
    rs, _ = synth.make_reads(g, 12, 9000, length_model="lognormal", seed=78)
    eng = hip.Engine(15, 5)
    p = eng.find_overlaps_and_create_piles(eng.upload(rs))
    data, off = p.piles()
    b, e, m, inv = p.trim_and_annotate(4)
    after, _ = p.piles()
    n_valid = 0
    for i in range(rs.n):
        want = data[int(off[i]):int(off[i + 1])].copy()
        wb, we, wm, winv = oracle.pile_trim_and_median(want, 4)
        assert (int(b[i]), int(e[i]), int(m[i]), bool(inv[i])) == (wb, we, wm, winv), i
        assert np.array_equal(after[int(off[i]):int(off[i + 1])], want), i
        n_valid += not winv
    assert 0.3 * rs.n < n_valid < rs.n            # both outcomes occur
    # idempotent on the trimmed data only where the region was terminated by the zeroed cells: run it again
    b2, e2, m2, inv2 = p.trim_and_annotate(4)
    ok = ~inv
    assert np.array_equal(b2[ok], b[ok]) and np.array_equal(e2[ok], e[ok]) and np.array_equal(m2[ok], m[ok])
    p.close()


def test_find_chimeric_regions_on_device_matches_oracle():
    """SURVEY 8(f) rank 1, third step of TrimAndAnnotatePiles: Pile::FindChimericRegions (FindSlopes(1.82), pit pairing,
    MergeRegions; pile.cc:176-187, :373-400, :403-600) on the trimmed coverage arrays in HBM vs the oracle's restatement,
    on the piles of a real pass over reads that include chimeras (two distant genome segments joined)."""
    g = synth.make_genome(200_000, seed=91)
    rs, truth = synth.make_reads(g, 25, 8000, seed=92)
    # make every 7th read chimeric: second half replaced by the first half of another read
    from raven_amd import seqio
    codes = [rs.codes(i) for i in range(rs.n)]
    for i in range(0, rs.n - 1, 7):
        codes[i] = np.concatenate([codes[i][:len(codes[i]) // 2], codes[(i + rs.n // 2) % rs.n][:4000]])
    rs = seqio.pack_reads(codes)
    eng = hip.Engine(15, 5)
    p = eng.find_overlaps_and_create_piles(eng.upload(rs))

    def check(inv):
        data, off = p.piles()
        got = p.find_chimeric_regions(inv)
        n_regions = 0
        for i in range(rs.n):
            if inv[i]:
                assert got[i].shape[0] == 0
                continue
            want = oracle.find_chimeric_regions(data[int(off[i]):int(off[i + 1])])
            assert got[i].shape == want.shape and np.array_equal(got[i], want), i
            n_regions += want.shape[0]
        return n_regions

    # on the raw coverage the junctions of the chimeric reads are pits down to ~0 ...
    assert check(np.zeros(rs.n, dtype=bool)) > 10
    # ... and in raven's order (FindValidRegion(4) first: the trim usually cuts a chimeric read at its junction)
    b, e, m, inv = p.trim_and_annotate(4)
    check(inv)
    p.close()
