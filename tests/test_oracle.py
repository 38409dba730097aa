"""CPU tests pinning the C++ oracle against definitional restatements (tests/refimpl.py).

The oracle's ram/biosoup parts are 'parity unpinned' against the real libraries (absent here, see
oracle/raven_oracle.cpp header); what CAN be pinned is pinned here: the sketch against the published
minimizer definition, the index against a dictionary, Filter against a numpy quantile, chains against
validity/optimality properties, AddLayers against per-cell counting, truncation against its contract,
overlaps against the simulator's ground truth.
"""
import numpy as np
import pytest

from oracle import oracle
from raven_amd import seqio, synth
from tests import refimpl


def _rand_reads(lengths, seed):
    rng = np.random.default_rng(seed)
    return seqio.pack_reads([rng.integers(0, 4, size=n, dtype=np.uint8) for n in lengths])


@pytest.mark.parametrize("k,w", [(15, 5), (5, 3), (19, 7), (31, 10), (15, 1), (11, 33)])
def test_sketch_matches_definition(k, w):
    rs = _rand_reads([0, 1, k - 1, k, k + w - 2, k + w - 1, k + w, 100, 777, 3000], seed=k * 100 + w)
    e = oracle.Engine(k, w)
    for i in range(rs.n):
        for mh in (False, True):
            v, o = e.sketch(rs, i, mh)
            bv, bo = refimpl.brute_sketch(rs.codes(i), k, w, int(rs.ids[i]), mh)
            assert np.array_equal(v, bv), (i, mh)
            assert np.array_equal(o, bo), (i, mh)


def test_sketch_low_complexity_and_palindromes():
    k, w = 15, 5
    reads = [np.zeros(500, np.uint8), np.tile(np.array([0, 3], np.uint8), 300),  # poly-A, (AT)n: palindromic k-mers
             np.tile(np.array([0, 1, 2, 3], np.uint8), 200), np.tile(np.array([1, 2], np.uint8), 250)]
    rs = seqio.pack_reads(reads)
    e = oracle.Engine(k, w)
    for i in range(rs.n):
        for mh in (False, True):
            v, o = e.sketch(rs, i, mh)
            bv, bo = refimpl.brute_sketch(rs.codes(i), k, w, i, mh)
            assert np.array_equal(v, bv) and np.array_equal(o, bo), (i, mh)


def test_sketch_lambda(lambda_reads):
    e = oracle.Engine(15, 5)
    for i in (0, 1, 2, 3, 100, 235):
        for mh in (False, True):
            v, o = e.sketch(lambda_reads, i, mh)
            bv, bo = refimpl.brute_sketch(lambda_reads.codes(i), 15, 5, i, mh)
            assert np.array_equal(v, bv) and np.array_equal(o, bo)
            if mh:
                assert v.shape[0] == min(bv.shape[0], int(lambda_reads.lengths[i]) // 15)


def test_index_and_filter(synth_small):
    _, rs, _ = synth_small
    e = oracle.Engine(15, 5)
    e.minimize(rs, 0, rs.n, False)
    table = {}
    for i in range(rs.n):
        v, o = refimpl.brute_sketch(rs.codes(i), 15, 5, i, False)
        for a, b in zip(v.tolist(), o.tolist()):
            table.setdefault(a, []).append(b)
    rng = np.random.default_rng(5)
    keys = list(table.keys())
    for key in rng.choice(len(keys), size=500, replace=False):
        val = keys[key]
        o, n = e.find(val)
        assert n == len(table[val]) and o.tolist() == table[val]
    assert e.find(12345678)[1] == len(table.get(12345678, []))
    counts = np.sort(np.array([len(x) for x in table.values()], dtype=np.uint32))
    for f in (0.001, 0.01, 0.5, 1.0):
        e.filter(f)
        assert e.occurrence == int(counts[min(int((1 - f) * counts.shape[0]), counts.shape[0] - 1)]) + 1
    e.filter(0)
    assert e.occurrence == 0xFFFFFFFF
    with pytest.raises(ValueError):
        e.filter(1.5)
    with pytest.raises(ValueError):
        e.filter(-0.1)


def _lis_valid(pos, idx_sorted, strand):
    lhs = (pos >> np.uint64(32)).astype(np.int64)[idx_sorted]
    rhs = (pos & np.uint64(0xFFFFFFFF)).astype(np.int64)[idx_sorted]
    return np.all(np.diff(lhs) > 0) and (np.all(np.diff(rhs) > 0) if strand else np.all(np.diff(rhs) < 0))


def test_chain_properties():
    """Chain on synthetic colinear matches: the overlap must span the planted chain and score must be the
    covered-bases count; noise matches off the diagonal band must not join."""
    e = oracle.Engine(15, 5)
    rng = np.random.default_rng(3)
    for strand in (1, 0):
        lhs = np.sort(rng.choice(9000, size=60, replace=False)).astype(np.uint64) + np.uint64(100)
        if strand:
            rhs = lhs + np.uint64(500) + rng.integers(0, 20, size=60).astype(np.uint64)
            rhs = np.maximum.accumulate(rhs) + np.arange(60, dtype=np.uint64)
            diag = rhs - lhs + np.uint64(3 << 30)
        else:
            rhs = np.uint64(20000) - lhs - rng.integers(0, 20, size=60).astype(np.uint64)
            rhs = np.minimum.accumulate(rhs) - np.arange(60, dtype=np.uint64)
            diag = rhs + lhs
        groups = ((np.uint64(7 << 1 | strand)) << np.uint64(32)) | diag
        positions = (lhs << np.uint64(32)) | rhs
        # noise on a far diagonal (fewer than 4 -> never an interval)
        ng = np.array([(7 << 1 | strand) << 32 | 12345] * 3, dtype=np.uint64)
        npos = np.array([(50 << 32) | 60, (80 << 32) | 95, (120 << 32) | 130], dtype=np.uint64)
        perm = rng.permutation(63)
        g = np.concatenate([groups, ng])[perm]
        p = np.concatenate([positions, npos])[perm]
        ovl = e.chain(3, g, p)
        assert ovl.shape[0] == 1
        o = ovl[0]
        assert o["lhs_id"] == 3 and o["rhs_id"] == 7 and o["strand"] == strand
        assert o["lhs_begin"] == lhs[0] and o["lhs_end"] == lhs[-1] + 15
        assert o["rhs_begin"] == rhs.min() and o["rhs_end"] == rhs.max() + 15
        # score = min over sides of the union length of the k-mer intervals
        def cov(x):
            x = np.sort(x.astype(np.int64))
            tot, b, en = 0, x[0], x[0] + 15
            for v in x[1:]:
                if v > en:
                    tot += en - b
                    b = v
                en = v + 15
            return tot + en - b
        assert o["score"] == min(cov(lhs), cov(rhs))


def test_chain_gap_split_and_thresholds():
    e = oracle.Engine(15, 5)
    # two colinear runs separated by > gap (10000) on lhs -> two overlaps; a run with < 100 covered bases -> dropped
    def run(start, n, step):
        lhs = np.uint64(start) + np.arange(n, dtype=np.uint64) * np.uint64(step)
        rhs = lhs + np.uint64(100)
        return lhs, rhs
    l1, r1 = run(100, 20, 30)
    l2, r2 = run(100 + 20 * 30 + 10500, 20, 30)
    lhs = np.concatenate([l1, l2])
    rhs = np.concatenate([r1, r2])
    g = (np.uint64(5 << 1 | 1) << np.uint64(32)) | (rhs - lhs + np.uint64(3 << 30))
    p = (lhs << np.uint64(32)) | rhs
    ovl = e.chain(1, g, p)
    assert ovl.shape[0] == 2
    assert ovl[0]["lhs_begin"] == 100 and ovl[1]["lhs_begin"] == l2[0]
    l3, r3 = run(100, 5, 10)  # covers only 55 bases < matches(100)
    g3 = (np.uint64(5 << 1 | 1) << np.uint64(32)) | (r3 - l3 + np.uint64(3 << 30))
    assert e.chain(1, g3, (l3 << np.uint64(32)) | r3).shape[0] == 0
    assert e.chain(1, g[:3], p[:3]).shape[0] == 0  # fewer than 4 matches


def test_add_layers_matches_counting():
    rng = np.random.default_rng(9)
    for trial in range(30):
        L = int(rng.integers(200, 20000))
        cells = L >> 4
        n = int(rng.integers(1, 120))
        ovl = np.zeros(n, oracle.OVERLAP_DTYPE)
        b = rng.integers(0, max(1, L - 120), size=n)
        en = np.minimum(L, b + rng.integers(100, L, size=n))
        side = rng.integers(0, 3, size=n)  # 0: lhs is the pile, 1: rhs is the pile, 2: unrelated
        ovl["lhs_id"] = np.where(side == 0, 42, 7)
        ovl["rhs_id"] = np.where(side == 1, 42, 8)
        ovl["lhs_begin"], ovl["lhs_end"] = np.where(side == 0, b, 5), np.where(side == 0, en, 500)
        ovl["rhs_begin"], ovl["rhs_end"] = np.where(side == 1, b, 9), np.where(side == 1, en, 900)
        data = rng.integers(0, 50, size=cells).astype(np.uint16)
        if trial % 5 == 0:
            data[:] = 65530  # saturation
        want = refimpl.naive_add_layers(data, 42, ovl)
        got = data.copy()
        oracle.pile_add_layers(got, 42, ovl)
        assert np.array_equal(got, want), trial


def test_truncate_contract():
    rng = np.random.default_rng(4)
    for n in (0, 5, 31, 32, 33, 100, 400):
        ovl = np.zeros(n, oracle.OVERLAP_DTYPE)
        ovl["lhs_begin"] = rng.integers(0, 100, size=n)
        ovl["lhs_end"] = ovl["lhs_begin"] + rng.integers(100, 130, size=n)  # many ties
        ovl["rhs_begin"] = 10
        ovl["rhs_end"] = 60
        ovl["score"] = np.arange(n)
        out = oracle.truncate(ovl, 32)
        if n < 32:
            assert np.array_equal(out, ovl)
            continue
        assert out.shape[0] == 32
        lens = (out["lhs_end"] - out["lhs_begin"]).astype(np.int64)
        assert np.all(np.diff(lens) <= 0)
        all_lens = np.sort((ovl["lhs_end"] - ovl["lhs_begin"]).astype(np.int64))[::-1]
        assert np.array_equal(lens, all_lens[:32])
        assert len(set(out["score"].tolist())) == 32  # a permutation subset, no duplicates


def test_pass1_structure_and_truth(synth_small):
    g, rs, truth = synth_small
    e = oracle.Engine(15, 5)
    r = e.find_overlaps_and_create_piles(rs, freq=0.001, kmax=32)
    off, ovl = r["overlap_offsets"], r["overlaps"]
    assert off[-1] == ovl.shape[0]
    start, end = truth["start"], truth["start"] + truth["src_len"]
    n_true = 0
    for i in range(rs.n):
        mine = ovl[int(off[i]): int(off[i + 1])]
        assert mine.shape[0] <= 32
        assert np.all(mine["lhs_id"] == i) and np.all(mine["rhs_id"] != i)
        assert np.all(mine["lhs_end"] <= rs.lengths[i]) and np.all(mine["lhs_begin"] < mine["lhs_end"])
        assert np.all(mine["rhs_end"] <= rs.lengths[mine["rhs_id"]])
        for o in mine:
            j = int(o["rhs_id"])
            inter = min(end[i], end[j]) - max(start[i], start[j])
            n_true += inter > 0
            # strand flag must agree with the simulator
            if inter > 200:
                assert bool(o["strand"]) == (truth["strand"][i] == truth["strand"][j])
    assert n_true >= 0.99 * ovl.shape[0]  # precision against ground truth
    # coverage sanity: mean pile coverage is a sizeable fraction of 20x
    assert 5 < r["pile_data"].mean() < 25
    # recall: most truly overlapping pairs (>= 2 kb shared) are found before truncation bites
    assert r["counters"]["overlaps"] > 0.7 * sum(
        1 for i in range(rs.n) for j in range(i + 1, rs.n)
        if min(end[i], end[j]) - max(start[i], start[j]) >= 2000)


def test_pass1_multibatch_consistency(synth_small):
    """Multi-threaded == single-threaded (submission-order merge); small flushes keep per-pile invariants."""
    _, rs, _ = synth_small
    a = oracle.Engine(15, 5).find_overlaps_and_create_piles(rs, threads=1)
    b = oracle.Engine(15, 5).find_overlaps_and_create_piles(rs, threads=4)
    assert np.array_equal(a["overlaps"], b["overlaps"]) and np.array_equal(a["pile_data"], b["pile_data"])
    c = oracle.Engine(15, 5).find_overlaps_and_create_piles(rs, flush_bases=rs.total_bases // 5)
    # coverage accumulation is independent of flush boundaries (AddLayers sees every overlap exactly once)
    assert np.array_equal(a["pile_data"], c["pile_data"])
    assert np.all(np.diff(c["overlap_offsets"].astype(np.int64)) <= 32)


def test_edit_distance():
    rng = np.random.default_rng(2)
    alphabet = np.frombuffer(b"ACGT", np.uint8)
    for _ in range(40):
        a = alphabet[rng.integers(0, 4, size=int(rng.integers(0, 60)))].tobytes()
        b = alphabet[rng.integers(0, 4, size=int(rng.integers(0, 60)))].tobytes()
        assert oracle.edit_distance(a, b) == refimpl.edit_distance(a, b)
    assert oracle.edit_distance(b"", b"ACGT") == 4
    assert oracle.edit_distance(b"ACGT", b"ACGT") == 0


def _brute_trim(data, coverage=4, min_cells=1260 >> 4):
    """Independent restatement of Pile::FindValidRegion + UpdateValidRegion + FindMedian (pile.cc:122-174) with numpy:
    maximal runs of cells >= coverage that are followed by a lower cell; first longest wins."""
    ge = np.concatenate([[False], data >= coverage, [False]])
    starts = np.nonzero(ge[1:] & ~ge[:-1])[0]
    ends = np.nonzero(~ge[1:] & ge[:-1])[0]
    runs = [(int(s), int(e)) for s, e in zip(starts, ends) if e < data.shape[0]]  # terminated inside the pile
    best = (0, 0)
    for s, e in runs:
        if e - s > best[1] - best[0]:
            best = (s, e)
    out = data.copy()
    if best[1] - best[0] < min_cells:
        return 0, data.shape[0], 0, True, out
    out[:best[0]] = 0
    out[best[1]:] = 0
    return best[0], best[1], int(np.sort(data[best[0]:best[1]])[(best[1] - best[0]) // 2]), False, out


def test_pile_trim_and_median_restatement():
    rng = np.random.default_rng(17)
    cases = [np.zeros(0, np.uint16), np.full(200, 9, np.uint16), np.array([9] * 100 + [0], np.uint16),
             np.array([0, 5, 5, 5, 1] + [9] * 100 + [0] + [7] * 90, np.uint16),
             np.array([9] * 77 + [0] + [9] * 78 + [0] + [9] * 78 + [0], np.uint16)]  # two equally long runs: first wins
    for _ in range(200):
        n = int(rng.integers(1, 700))
        d = rng.integers(0, 12, size=n).astype(np.uint16)
        # plant plateaus so that long runs exist
        for _ in range(int(rng.integers(0, 4))):
            a = int(rng.integers(0, n))
            d[a:a + int(rng.integers(1, 300))] += np.uint16(rng.integers(4, 4000))
        cases.append(d)
    n_valid = 0
    for d in cases:
        got = d.copy()
        b, e, m, inv = oracle.pile_trim_and_median(got)
        wb, we, wm, winv, wout = _brute_trim(d)
        assert (b, e, m, inv) == (wb, we, wm, winv), (d.tolist()[:20], (b, e, m, inv), (wb, we, wm, winv))
        assert np.array_equal(got, wout)
        n_valid += not inv
    assert n_valid > 20
    b, e, m, inv = oracle.pile_trim_and_median(cases[4].copy())
    assert (b, e, inv) == (78, 156, False)


def test_banded_alignment_path_is_the_full_matrix_path():
    """The oracle's NW path runs in an Ukkonen band that doubles until it holds the distance; path (hence breakpoints) and
    distance must be the full matrix's, ties included: random pairs from identical to unrelated, lengths that differ,
    indel bursts that push the path to the band's edge, homopolymers (ties everywhere)."""
    rng = np.random.default_rng(77)

    def mutate(x, sub, ins, dele):
        out = []
        for c in x:
            u = rng.random()
            if u < dele:
                continue
            out.append((int(c) + int(rng.integers(1, 4))) & 3 if u < dele + sub else int(c))
            if rng.random() < ins:
                out.append(int(rng.integers(0, 4)))
        return np.asarray(out, dtype=np.uint8)

    cases = []
    for trial in range(60):
        n = int(rng.integers(1, 1500))
        t = rng.integers(0, 4, n).astype(np.uint8)
        e = float(rng.choice([0.0, 0.01, 0.1, 0.3]))
        q = mutate(t, e, e / 2, e / 2)
        if trial % 7 == 0:  # an indel burst
            cut = int(rng.integers(0, max(1, len(q))))
            q = np.concatenate([q[:cut], rng.integers(0, 4, int(rng.integers(1, 120))).astype(np.uint8), q[cut:]])
        if trial % 11 == 0:
            q = q[: max(1, len(q) // 2)]
        if len(q) == 0:
            q = np.zeros(1, dtype=np.uint8)
        cases.append((q, t))
    cases.append((np.zeros(700, dtype=np.uint8), np.zeros(640, dtype=np.uint8)))  # homopolymers
    cases.append((rng.integers(0, 4, 900).astype(np.uint8), rng.integers(0, 4, 800).astype(np.uint8)))  # unrelated
    try:
        for q, t in cases:
            oracle.nw_full_matrix(False)
            a, da = oracle.nw_breakpoints(q, t, 0, 3, 100)
            oracle.nw_full_matrix(True)
            b, db = oracle.nw_breakpoints(q, t, 0, 3, 100)
            assert da == db and np.array_equal(a, b), (len(q), len(t))
    finally:
        oracle.nw_full_matrix(False)
