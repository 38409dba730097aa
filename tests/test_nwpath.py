"""The alignment-path stage of a polishing round (raven_amd/csrc/nwpath.h: banded Myers forward sweep with stored
vertical deltas + traceback + racon's find_breaking_points) against the oracle's plain-DP path and breakpoints.

No GPU needed: rvn_test_nw_breakpoints drives the SAME __host__ __device__ code the kernels execute — the forward
sweep's per-lane step for 64 emulated lanes (nwsweep.h: ring reuse, systolic carries through the emulated shuffles, the
hs / checkpoint stores) and the one-lane traceback that recomputes one block at a time from them (nwtrace.h) — so the
arithmetic, the band geometry, the store layout and the tie rule are all pinned here;
the GPU tests (tests/test_gpu_polish.py) then only have to show that the wave executes it the same way."""
import numpy as np
import pytest

from oracle import oracle
from raven_amd import hip, seqio, synth


def _pack(codes):
    rs = seqio.pack_reads([np.asarray(codes, dtype=np.uint8)])
    return np.concatenate([rs.packed, np.zeros(2, np.uint64)])


def _oriented(read_codes, rc):
    return (3 - read_codes[::-1]) if rc else read_codes


def _check(target, read, t_begin, n, q_begin, m, rc, w, k=64, force_r=0, group_lanes=0):
    """read: codes as stored (original orientation); the alignment uses its reverse complement when rc."""
    tw, rw = _pack(target), _pack(read)
    recs, dist, band, status = hip.test_nw_breakpoints(tw, len(target), rw, len(read), t_begin, n, q_begin, m, rc, w,
                                                       k=k, force_r=force_r, group_lanes=group_lanes)
    assert status == 0
    if group_lanes:  # the walk by a group of lanes: the records of the walk by one lane, byte for byte
        recs1, dist1, _, status1 = hip.test_nw_breakpoints(tw, len(target), rw, len(read), t_begin, n, q_begin, m, rc, w,
                                                           k=k, force_r=force_r)
        assert status1 == 0 and dist1 == dist and recs.tobytes() == recs1.tobytes()
    rq = _oriented(np.asarray(read, dtype=np.uint8), rc)
    want, want_dist = oracle.nw_breakpoints(rq[q_begin:q_begin + m], np.asarray(target[t_begin:t_begin + n], np.uint8),
                                            q_begin, t_begin, w)
    assert dist == want_dist
    got = []
    for x, r in enumerate(recs):
        if r["first_t"] == 0xFFFFFFFF:
            continue
        assert r["first_t"] // w == t_begin // w + x  # record x belongs to window x of the span
        got.append((int(r["first_t"]), int(r["first_q"])))
        got.append((int(r["last_t"]), int(r["last_q"])))
        # band guide: monotone read offsets inside the piece, at the fixed target positions the path covers
        g = [int(v) for v in r["grid"] if v != 0xFFFF]
        assert g == sorted(g) and all(v <= r["last_q"] - r["first_q"] for v in g)
    assert got == [tuple(int(v) for v in p) for p in want]
    return dist, band


def _noisy_pair(rng, n, sub, ins, dele):
    t = rng.integers(0, 4, size=n, dtype=np.uint8)
    q = synth.mutate(rng, t, sub, ins, dele)
    return t, q


@pytest.mark.parametrize("rc", [0, 1])
def test_breakpoints_match_oracle_ont_like(rc):
    rng = np.random.default_rng(100 + rc)
    for trial in range(6):
        n = int(rng.integers(700, 3000))
        t, q = _noisy_pair(rng, n, 0.04, 0.03, 0.03)
        # embed the spans inside longer sequences at unaligned offsets
        tl, ql = int(rng.integers(0, 700)), int(rng.integers(0, 90))
        target = np.concatenate([rng.integers(0, 4, tl, dtype=np.uint8), t, rng.integers(0, 4, 77, dtype=np.uint8)])
        read_o = np.concatenate([rng.integers(0, 4, ql, dtype=np.uint8), q, rng.integers(0, 4, 33, dtype=np.uint8)])
        read = _oriented(read_o, rc)  # stored orientation: reverse complement when the overlap is on the other strand
        dist, band = _check(target, read, tl, len(t), ql, len(q), rc, 500)
        assert band[0] >= dist and dist > 0.05 * n


def test_low_error_and_identical():
    rng = np.random.default_rng(7)
    t, q = _noisy_pair(rng, 4000, 0.001, 0.002, 0.002)
    _check(t, q, 0, len(t), 0, len(q), 0, 500)
    t = rng.integers(0, 4, 1500, dtype=np.uint8)
    dist, _ = _check(t, t.copy(), 0, 1500, 0, 1500, 0, 500)
    assert dist == 0


def test_band_doubling_and_blocks_per_lane():
    """A first threshold far below the distance is doubled until exact; every blocks-per-lane variant of the forward
    kernel (R = 1, 2, 4, 8) gives the same path."""
    rng = np.random.default_rng(21)
    t, q = _noisy_pair(rng, 2600, 0.06, 0.05, 0.05)
    d1, b1 = _check(t, q, 0, len(t), 0, len(q), 0, 500, k=8)
    assert b1[0] >= d1 and b1[0] <= max(4 * d1, 64)
    for R in (2, 4, 8):
        d, b = _check(t, q, 0, len(t), 0, len(q), 0, 500, k=8, force_r=R)
        assert d == d1 and b[2] == R


def test_lane_group_variants_give_the_same_breakpoints():
    """The narrow variants of the sweep (rings of at most 4 / 8 / 16 / 32 lanes sharing a wave) stepped on the CPU: same
    distance, same breakpoints as the oracle; a band wider than the ring is refused, not truncated."""
    rng = np.random.default_rng(44)
    for trial in range(4):
        n = int(rng.integers(500, 2600))
        t, q = _noisy_pair(rng, n, 0.04, 0.03, 0.03)
        rc = trial & 1
        read = _oriented(q, rc)
        for g in (16, 32, 64):
            d, band = _check(t, read, 0, len(t), 0, len(q), rc, 500, k=32, force_r=-g)
            assert band[1] <= g and band[0] >= d
    t, q = _noisy_pair(rng, 3000, 0.002, 0.002, 0.002)  # HiFi-like: the smallest ring
    _, band = _check(t, q, 0, len(t), 0, len(q), 0, 500, k=16, force_r=-4)
    assert band[1] <= 4
    for n, m in [(1, 1), (1, 7), (9, 1), (64, 64), (65, 63), (300, 250)]:
        tt = rng.integers(0, 4, n, dtype=np.uint8)
        qq = rng.integers(0, 4, m, dtype=np.uint8)
        _check(tt, qq, 0, n, 0, m, 0, 50, k=4, force_r=-8)
    t, q = _noisy_pair(rng, 2500, 0.08, 0.06, 0.06)  # distance ~450: beyond a ring of 4 lanes
    with pytest.raises(ValueError):
        _check(t, q, 0, len(t), 0, len(q), 0, 500, k=600, force_r=-4)


def test_length_difference_and_indel_bursts():
    rng = np.random.default_rng(33)
    t = rng.integers(0, 4, 2500, dtype=np.uint8)
    q = np.concatenate([t[:800], rng.integers(0, 4, 300, dtype=np.uint8), t[800:1700], t[1950:]])  # +300 / -250
    _check(t, q, 0, len(t), 0, len(q), 0, 500)
    _check(q, t, 0, len(q), 0, len(t), 0, 500)
    # very different lengths: the band is one-sided
    _check(t[:400], np.concatenate([t[:400], rng.integers(0, 4, 900, dtype=np.uint8)]), 0, 400, 0, 1300, 0, 100)
    _check(np.concatenate([t[:400], rng.integers(0, 4, 900, dtype=np.uint8)]), t[:400], 0, 1300, 0, 400, 0, 100)


def test_small_and_degenerate_spans():
    rng = np.random.default_rng(5)
    for n, m in [(1, 1), (1, 7), (9, 1), (63, 64), (64, 64), (65, 63), (128, 129), (5, 200)]:
        t = rng.integers(0, 4, n, dtype=np.uint8)
        q = rng.integers(0, 4, m, dtype=np.uint8)
        _check(t, q, 0, n, 0, m, 0, 50)
    h = np.zeros(300, dtype=np.uint8)  # homopolymers: every path is optimal, the tie rule decides
    _check(h, h[:250], 0, 300, 0, 250, 0, 100)
    _check(h[:250], h, 0, 250, 0, 300, 0, 100)


def test_unrelated_sequences_and_window_sizes():
    rng = np.random.default_rng(9)
    t = rng.integers(0, 4, 900, dtype=np.uint8)
    q = rng.integers(0, 4, 1000, dtype=np.uint8)
    _check(t, q, 0, 900, 0, 1000, 0, 500)
    t, q = _noisy_pair(rng, 1800, 0.03, 0.03, 0.03)
    for w in (37, 64, 500, 1023):
        _check(t, q, 0, len(t), 0, len(q), 0, w)
    for tb in (0, 1, 499, 500, 501):  # spans starting at / around a window boundary
        target = np.concatenate([rng.integers(0, 4, tb, dtype=np.uint8), t])
        _check(target, q, tb, len(t), 0, len(q), 0, 500)


def test_real_reads_lambda(lambda_reads, lambda_genome):
    """The reference's own test data (RavenTest/data): the first reads against the slices of NC_001416 they map to."""
    eng = oracle.Engine(15, 5)
    eng.minimize(lambda_genome, minhash=False)
    eng.filter(0.001)
    done = 0
    for r in range(lambda_reads.n):
        if lambda_reads.lengths[r] > 4200:
            continue
        ovl = eng.map(lambda_reads, r, avoid_equal=False, avoid_symmetric=False, minhash=False)["overlaps"]
        if len(ovl) == 0:
            continue
        o = max(ovl, key=lambda x: max(int(x["lhs_end"]) - int(x["lhs_begin"]), int(x["rhs_end"]) - int(x["rhs_begin"])))
        read = lambda_reads.codes(r)
        target = lambda_genome.codes(0)
        rc = int(o["strand"]) == 0
        qlen = len(read)
        q_begin = qlen - int(o["lhs_end"]) if rc else int(o["lhs_begin"])
        _check(target, read, int(o["rhs_begin"]), int(o["rhs_end"]) - int(o["rhs_begin"]), q_begin,
               int(o["lhs_end"]) - int(o["lhs_begin"]), int(rc), 500)
        done += 1
        if done == 3:
            break
    assert done == 3


def _reference_records(tq, tt, q_begin, t_begin, w):
    """Independent statement of what the walker records, from a plain full-matrix NW (unit costs) walked back with
    racon's CIGAR rule (match => diagonal; else substitution, then read-only base, then target-only base): per window of
    the target the first / last match pair and, at the 8 grid positions start + g w / 8, the oriented read position at the
    moment the path consumed that target base (the next read position for a target-only step)."""
    n, m = len(tt), len(tq)
    D = np.zeros((n + 1, m + 1), dtype=np.int64)
    D[:, 0] = np.arange(n + 1)
    D[0, :] = np.arange(m + 1)
    for i in range(1, n + 1):
        row, prev = D[i], D[i - 1]
        ti = tt[i - 1]
        for j in range(1, m + 1):
            row[j] = min(prev[j - 1] + (0 if ti == tq[j - 1] else 1), row[j - 1] + 1, prev[j] + 1)
    wins = {}

    def consume(t, q, is_match):
        rec = wins.setdefault(t // w, dict(first=None, last=None, grid={}))
        start = (t // w) * w
        for g in range(8):  # windows shorter than 8 bases: several g share a position, the largest one holds the sample
            if t == start + (g * w) // 8 and (g == 7 or ((g + 1) * w) // 8 != (g * w) // 8):
                rec["grid"][g] = q
        if is_match:
            if rec["last"] is None:
                rec["last"] = (t + 1, q + 1)
            rec["first"] = (t, q)

    i, j = n, m
    while i > 0 and j > 0:
        if tt[i - 1] == tq[j - 1] or D[i - 1, j - 1] + 1 == D[i, j]:
            consume(t_begin + i - 1, q_begin + j - 1, True)
            i, j = i - 1, j - 1
        elif D[i, j - 1] + 1 == D[i, j]:
            j -= 1
        else:
            consume(t_begin + i - 1, q_begin + j, False)
            i -= 1
    while i > 0:
        consume(t_begin + i - 1, q_begin, False)
        i -= 1
    return int(D[n, m]), wins


@pytest.mark.parametrize("w", [7, 50, 64, 500])
def test_window_records_and_grid_samples_equal_an_independent_traceback(w):
    """The walk takes whole runs of matches at once and applies the window bookkeeping in closed form: every field of the
    records, grid samples included, must equal a base-by-base traceback written independently in Python."""
    rng = np.random.default_rng(900 + w)
    for trial in range(5):
        n = int(rng.integers(60, 330))
        t, q = _noisy_pair(rng, n, 0.05, 0.04, 0.04)
        if trial == 3:  # a long run of identical bases and a long gap
            q = np.concatenate([t[:n // 3], t[n // 3 + 25:]])
        rc = trial & 1
        tl, ql = int(rng.integers(0, 3 * w + 5)), int(rng.integers(0, 40))
        target = np.concatenate([rng.integers(0, 4, tl, dtype=np.uint8), t, rng.integers(0, 4, 9, dtype=np.uint8)])
        read_o = np.concatenate([rng.integers(0, 4, ql, dtype=np.uint8), q, rng.integers(0, 4, 5, dtype=np.uint8)])
        read = _oriented(read_o, rc)
        for force_r in (0, -8, 2):  # the narrowest variant that fits, a ring of <= 8 lanes, two blocks per lane
            recs, dist, band, status = hip.test_nw_breakpoints(_pack(target), len(target), _pack(read), len(read), tl, len(t),
                                                               ql, len(q), rc, w, k=16, force_r=force_r)
            assert status == 0
            want_dist, wins = _reference_records(q, t, ql, tl, w)
            assert dist == want_dist
            for x, r in enumerate(recs):
                ref = wins.get(tl // w + x)
                if ref is None or ref["first"] is None:
                    assert r["first_t"] == 0xFFFFFFFF
                    continue
                assert (int(r["first_t"]), int(r["first_q"])) == ref["first"]
                assert (int(r["last_t"]), int(r["last_q"])) == ref["last"]
                span = ref["last"][1] - ref["first"][1]
                for g in range(8):
                    if g in ref["grid"]:
                        off = min(max(ref["grid"][g] - ref["first"][1], 0), span, 0xFFFE)
                        assert int(r["grid"][g]) == off, (trial, x, g)
                    else:
                        assert int(r["grid"][g]) == 0xFFFF


@pytest.mark.parametrize("gl", [1, 4, 16, 64])
def test_group_walk_equals_the_lane_walk(gl):
    """nwtrace.h's walk by a group of lanes per alignment (the strips along the predicted path recomputed side by side, one
    walker stepping through them), its phases stepped lane by lane: the records of the one-lane walk byte for byte and the
    oracle's breakpoints — where the prediction holds (few batches) and where it cannot (bursts, unrelated sequences,
    one-sided bands, degenerate spans), on every ring layout of the sweep.  gl = 1: the one-lane walk with strips of sixteen
    kept columns (what the kernel runs where the strips live in LDS) against the one with whole strips."""
    rng = np.random.default_rng(600 + gl)
    for trial in range(4):  # ONT-like, both strands, embedded at unaligned offsets
        n = int(rng.integers(900, 3200))
        t, q = _noisy_pair(rng, n, 0.04, 0.03, 0.03)
        rc = trial & 1
        tl, ql = int(rng.integers(0, 700)), int(rng.integers(0, 90))
        target = np.concatenate([rng.integers(0, 4, tl, dtype=np.uint8), t, rng.integers(0, 4, 77, dtype=np.uint8)])
        read_o = np.concatenate([rng.integers(0, 4, ql, dtype=np.uint8), q, rng.integers(0, 4, 33, dtype=np.uint8)])
        dist, band = _check(target, _oriented(read_o, rc), tl, len(t), ql, len(q), rc, 500, group_lanes=gl)
        # the prediction serves: a batch walks through several strips (a strip is at most 32 columns and 64 rows)
        strips_at_least = len(q) // 32
        assert gl == 1 or band[3] <= max(2, strips_at_least // min(gl // 2, 4)), (band, len(q))
    t, q = _noisy_pair(rng, 3000, 0.002, 0.002, 0.002)  # HiFi-like, the narrowest ring
    _check(t, q, 0, len(t), 0, len(q), 0, 500, k=16, force_r=-4, group_lanes=gl)
    t, q = _noisy_pair(rng, 2200, 0.04, 0.03, 0.03)
    for R in (1, 2, 4, 8):  # several blocks per lane: the strips of a super-block share a column phase
        _check(t, q, 0, len(t), 0, len(q), 0, 500, k=64, force_r=R, group_lanes=gl)
    for g in (16, 32):
        rc = 1 if g == 32 else 0
        _check(t, _oriented(q, rc), 0, len(t), 0, len(q), rc, 500, k=32, force_r=-g, group_lanes=gl)
    # where the straight line is a bad guess
    t = rng.integers(0, 4, 2500, dtype=np.uint8)
    q = np.concatenate([t[:800], rng.integers(0, 4, 300, dtype=np.uint8), t[800:1700], t[1950:]])  # +300 / -250
    _check(t, q, 0, len(t), 0, len(q), 0, 500, group_lanes=gl)
    _check(q, t, 0, len(q), 0, len(t), 0, 500, group_lanes=gl)
    _check(t[:400], np.concatenate([t[:400], rng.integers(0, 4, 900, dtype=np.uint8)]), 0, 400, 0, 1300, 0, 100, group_lanes=gl)
    _check(np.concatenate([t[:400], rng.integers(0, 4, 900, dtype=np.uint8)]), t[:400], 0, 1300, 0, 400, 0, 100, group_lanes=gl)
    _check(rng.integers(0, 4, 900, dtype=np.uint8), rng.integers(0, 4, 1000, dtype=np.uint8), 0, 900, 0, 1000, 0, 500, group_lanes=gl)
    for n, m in [(1, 1), (1, 7), (9, 1), (63, 64), (64, 64), (65, 63), (128, 129), (5, 200), (200, 5)]:
        _check(rng.integers(0, 4, n, dtype=np.uint8), rng.integers(0, 4, m, dtype=np.uint8), 0, n, 0, m, 0, 50, group_lanes=gl)
    h = np.zeros(300, dtype=np.uint8)  # homopolymers: the tie rule decides every step
    _check(h, h[:250], 0, 300, 0, 250, 0, 100, group_lanes=gl)
    _check(h[:250], h, 0, 250, 0, 300, 0, 100, group_lanes=gl)
    t, q = _noisy_pair(rng, 1800, 0.03, 0.03, 0.03)
    for w in (37, 500, 1023):
        _check(t, q, 0, len(t), 0, len(q), 0, w, group_lanes=gl)
