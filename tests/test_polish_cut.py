"""Host logic of the polishing front end (raven_amd/csrc/polish_cut.h, no GPU needed): where a read is cut at a
window boundary of its target.  Reads are simulated with a known base-to-base mapping, anchors are the exact k-mer
matches along it (what the mapping stage hands over)."""
import numpy as np
import pytest

from raven_amd import hip

K = 15


def simulate(rng, target, sub, ins, dele):
    """read + for every read base the target position it came from (-1 = inserted) + per target base the read index
    of its copy (-1 = deleted / substituted keeps its index)."""
    read, src = [], []
    t2q = np.full(target.shape[0], -1, dtype=np.int64)
    for i, c in enumerate(target):
        u = rng.random()
        if u < dele:
            pass
        else:
            if u < dele + sub:
                c = (c + rng.integers(1, 4)) & 3
            t2q[i] = len(read)
            read.append(int(c))
            src.append(i)
        if rng.random() < ins:
            read.append(int(rng.integers(0, 4)))
            src.append(-1)
    return np.array(read, dtype=np.uint8), np.array(src), t2q


def anchors_of(target, read, t2q, k, stride=7):
    """Exact k-mer matches on the true diagonal, thinned like minimizers are."""
    out = []
    for t in range(0, target.shape[0] - k, 1):
        q = t2q[t]
        if q < 0 or q + k > read.shape[0]:
            continue
        if np.array_equal(t2q[t:t + k], np.arange(q, q + k)) and np.array_equal(target[t:t + k], read[q:q + k]):
            if not out or t >= out[-1][0] + stride:
                out.append((t, int(q)))
    return out


def test_error_free_read_is_cut_exactly():
    rng = np.random.default_rng(1)
    target = rng.integers(0, 4, size=3000, dtype=np.uint8)
    read = target[200:2800].copy()
    an = [(t, t - 200) for t in range(200, 2700, 40)]
    for B in (500, 1000, 1013, 2500):
        ql, tl, qr, tr, n_nw = hip.test_window_cut(target, read, [a[0] for a in an], [a[1] for a in an], K, B)
        assert (ql, tl, qr, tr, n_nw) == (B - 200, B, B - 200, B, 0)


@pytest.mark.parametrize("rates", [(0.01, 0.005, 0.005), (0.04, 0.03, 0.03), (0.06, 0.05, 0.05)])
def test_cuts_follow_the_true_alignment(rates):
    rng = np.random.default_rng(int(rates[0] * 1000))
    checked = nw_used = 0
    for _ in range(6):
        target = rng.integers(0, 4, size=6000, dtype=np.uint8)
        read, src, t2q = simulate(rng, target, *rates)
        an = anchors_of(target, read, t2q, K)
        assert len(an) > 20
        at, aq = [a[0] for a in an], [a[1] for a in an]
        for B in range(500, 5600, 500):
            if not (at[0] <= B < at[-1] + K):
                continue
            ql, tl, qr, tr, n_nw = hip.test_window_cut(target, read, at, aq, K, B)
            nw_used += n_nw
            checked += 1
            # ordering: left piece ends at/before B, right piece begins at/after it, pieces do not overlap
            assert tl <= B <= tr and ql <= qr and tl <= tr
            assert B - tl <= 40 and tr - B <= 40          # only the indel bases at the cut are given up
            # both cut points lie on (or within a couple of bases of) the true base-to-base mapping
            for tpos, qpos in ((tl - 1, ql - 1), (tr, qr)):
                near = [t2q[x] for x in range(max(0, tpos - 3), min(target.shape[0], tpos + 4)) if t2q[x] >= 0]
                assert near and min(abs(qpos - v) for v in near) <= 4, (B, tpos, qpos, near)
    assert checked > 40
    if rates[0] >= 0.04:
        assert nw_used > 0  # clusters of errors do reach the residual NW


def test_boundary_inside_an_indel_gives_the_bases_to_neither_piece():
    rng = np.random.default_rng(5)
    target = rng.integers(0, 4, size=1200, dtype=np.uint8)
    # read = target with bases 598..603 deleted: boundary 600 falls into the deletion
    read = np.concatenate([target[:598], target[604:]])
    an = [(t, t) for t in range(100, 560, 30)] + [(t, t - 6) for t in range(640, 1100, 30)]
    ql, tl, qr, tr, _ = hip.test_window_cut(target, read, [a[0] for a in an], [a[1] for a in an], K, 600)
    assert tl <= 600 <= tr and ql <= qr
    assert tr - tl >= 6 and qr - ql <= 2  # the deleted target bases are skipped, (almost) no read base is
    # read with 5 extra bases inserted right at the boundary
    read2 = np.concatenate([target[:600], (target[600:605] + 1) & 3, target[600:]])
    an2 = [(t, t) for t in range(100, 560, 30)] + [(t, t + 5) for t in range(640, 1100, 30)]
    ql, tl, qr, tr, _ = hip.test_window_cut(target, read2, [a[0] for a in an2], [a[1] for a in an2], K, 600)
    assert tl <= 600 <= tr and tr - tl <= 2 and 3 <= qr - ql <= 7  # the inserted read bases belong to neither piece


def test_invalid_arguments():
    t = np.zeros(100, np.uint8)
    with pytest.raises(ValueError):
        hip.test_window_cut(t, t, [10], [10], K, 20)            # fewer than two anchors
    with pytest.raises(ValueError):
        hip.test_window_cut(t, t, [10, 40], [10, 40], K, 80)    # boundary outside the chain
