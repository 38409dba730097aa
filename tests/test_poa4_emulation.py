"""The rows-on-lanes banded POA kernel (raven_amd/csrc/poa4.hip) stepped through on the CPU: the kernel source is written
against sv:: (csrc/simt.h) and runs unchanged under the 64-fibre wavefront emulator (csrc/simt_emu.hip), so the whole
kernel — per-layer row descriptors, the systolic NW (a graph row per lane, two columns per step), lock-step tracebacks
through the time-major backpointer stream, graph update, consensus — is compared with the POA oracle (racon
Window::GenerateConsensus over spoa) here, without a GPU.  First attempt of the escalation chain only: a window whose
alignment touches the 32-column band, or whose graph is beyond the kernel's limits (an in-edge longer than the LDS ring,
more than eight in-edges), comes back flagged (status 8) and is not compared.
The GPU side of the same comparison is tests/test_gpu_poa.py (mode 9)."""
import numpy as np
import pytest

from oracle import oracle
from raven_amd import hip


# 4 = the per-round kernels (graph side / alignment side), 5 = the persistent kernel (one launch, a wave carries a group of four
# windows through all its layers): the same phase functions, scheduled differently — every test runs under both
VARIANT = [4]


@pytest.fixture(autouse=True, params=[4, 5], ids=["per_round", "persistent"])
def _variant(request):
    VARIANT[0] = request.param
    yield
    VARIANT[0] = 4


def _mutate(rng, codes, sub, ins, dele):
    out = []
    for c in codes:
        u = rng.random()
        if u < dele:
            continue
        if u < dele + sub:
            c = (c + rng.integers(1, 4)) & 3
        out.append(int(c))
        if rng.random() < ins:
            out.append(int(rng.integers(0, 4)))
    return np.array(out, dtype=np.uint8)


def _window(rng, length, n_reads, err=(0.05, 0.04, 0.04), partial=0.0, qual=False):
    truth = rng.integers(0, 4, size=length, dtype=np.uint8)
    bb = _mutate(rng, truth, 0.03, 0.02, 0.02)
    layers, begins, ends = [bb], [0], [len(bb) - 1]
    quals = [np.full(len(bb), 33, np.uint8)] if qual else None
    for _ in range(n_reads):
        if rng.random() < partial:
            b = int(rng.integers(0, length // 2))
            e = int(rng.integers(b + length // 4, length))
        else:
            b, e = 0, length
        piece = _mutate(rng, truth[b:e], *err)
        if len(piece) < 2:
            continue
        layers.append(piece)
        bb_b = min(len(bb) - 2, int(b * len(bb) / length))
        bb_e = min(len(bb) - 1, max(bb_b + 1, int(e * len(bb) / length) - 1))
        begins.append(bb_b)
        ends.append(bb_e)
        if qual:
            quals.append((33 + rng.integers(5, 40, size=len(piece))).astype(np.uint8))
    return dict(layers=layers, begins=begins, ends=ends, quals=quals)


def _oracle(w, trim=True):
    return oracle.poa_window(w["layers"], begins=w.get("begins"), ends=w.get("ends"), quals=w.get("quals"), trim=trim)[0]


def _compare(wins, min_polished, **kw):
    cons, status = hip.poa_banded_emulate(wins, variant=VARIANT[0], **kw)
    polished = 0
    for i, (w, c, st) in enumerate(zip(wins, cons, status)):
        if (int(st) & 0xFF) == 1:
            polished += 1
            assert np.array_equal(c, _oracle(w, trim=kw.get("trim", True))), (i, kw)
        else:
            assert (int(st) & 0xFF) in (0, 8), st
    assert polished >= min_polished, status
    return cons, status


def test_simple_windows():
    rng = np.random.default_rng(1)
    truth = rng.integers(0, 4, size=150, dtype=np.uint8)
    bb = _mutate(rng, truth, 0.03, 0.02, 0.02)
    wins = [
        dict(layers=[bb] + [truth.copy() for _ in range(6)]),      # error-free layers fix the backbone
        dict(layers=[bb, truth.copy()]),                            # < 3 sequences: backbone back
        dict(layers=[bb]),
        dict(layers=[truth.copy()] + [truth[30:120].copy() for _ in range(8)], begins=[0] + [30] * 8,
             ends=[149] + [119] * 8),                               # trimming of thin ends
        dict(layers=[bb] + [truth.copy() for _ in range(3)]),       # a fifth window: the wave's second batch of four
    ]
    cons, status = hip.poa_banded_emulate(wins, variant=VARIANT[0])
    assert status.tolist() == [1, 0, 0, 1, 1]
    assert np.array_equal(cons[0], truth)
    assert np.array_equal(cons[1], bb) and np.array_equal(cons[2], bb)
    assert np.array_equal(cons[3], truth[30:120])
    for w, c in zip(wins, cons):
        assert np.array_equal(c, _oracle(w))
    cons_nt, _ = hip.poa_banded_emulate(wins[3:4], trim=False, variant=VARIANT[0])
    assert np.array_equal(cons_nt[0], _oracle(wins[3], trim=False))


def test_noisy_windows_ragged_groups():
    """Windows of different sizes, layer counts, partial layers and qualities share a wave: the four groups run out of
    rows, steps, layers and traceback steps at different times."""
    rng = np.random.default_rng(5)
    wins = []
    for i in range(14):
        wins.append(_window(rng, int(rng.integers(70, 260)), int(rng.integers(3, 14)), partial=0.3 if i % 2 else 0.0,
                            qual=(i % 3 == 0)))
    _compare(wins, min_polished=11)


def test_window_sized_like_racon():
    """500-base windows with 30 layers (the shape a polishing round produces), two of them with partial layers."""
    rng = np.random.default_rng(11)
    wins = [_window(rng, 500 + 20 * i, 30, partial=0.25 if i % 2 else 0.0, qual=(i == 0)) for i in range(4)]
    _compare(wins, min_polished=3)


def test_short_and_tiny_windows():
    """Layers shorter than the band (every row holds the whole layer), windows shorter than a block of 16 rows."""
    rng = np.random.default_rng(9)
    wins = [_window(rng, n, 6, err=(0.03, 0.02, 0.02)) for n in (5, 12, 17, 30, 33, 47, 64, 65)]
    _compare(wins, min_polished=6)


def test_long_private_insertion_is_flagged_or_exact():
    """A long private insertion in half of the reads: an in-edge that spans more rows than the LDS ring keeps (or an
    alignment that leaves the 32-column band) sends the window on to the 64-column kernel; whatever is polished here is
    exact."""
    rng = np.random.default_rng(3)
    wins = []
    for _ in range(4):
        truth = rng.integers(0, 4, size=220, dtype=np.uint8)
        layers = [_mutate(rng, truth, 0.03, 0.02, 0.02)]
        for r in range(10):
            t = truth
            if r % 2 == 0:
                pos = 100 + int(rng.integers(0, 5))
                t = np.concatenate([truth[:pos], rng.integers(0, 4, size=int(rng.integers(18, 30)), dtype=np.uint8), truth[pos:]])
            layers.append(_mutate(rng, t, 0.03, 0.02, 0.02))
        wins.append(dict(layers=layers))
    _compare(wins, min_polished=0)


def test_scoring_parameters():
    rng = np.random.default_rng(21)
    wins = [_window(rng, 120, 8) for _ in range(4)]
    cons, status = hip.poa_banded_emulate(wins, m=5, n=-4, g=-8, variant=VARIANT[0])
    polished = 0
    for w, c, st in zip(wins, cons, status):
        if (int(st) & 0xFF) == 1:
            polished += 1
            o = oracle.poa_window(w["layers"], begins=w["begins"], ends=w["ends"], quals=w["quals"], m=5, n=-4, g=-8)[0]
            assert np.array_equal(c, o)
    assert polished >= 3


def test_nodes_with_many_in_edges():
    """Insertions of every letter (and of two letters) in front of the same backbone position give that node five and
    more in-edges: in-edges 0..7 sit in the row descriptor (the traceback resolves 6 and 7 through the graph)."""
    rng = np.random.default_rng(17)
    truth = rng.integers(0, 4, size=160, dtype=np.uint8)
    layers = [truth.copy()]
    for rep in range(2):
        for letter in range(4):
            layers.append(np.concatenate([truth[:70], np.array([letter], np.uint8), truth[70:]]))
            layers.append(np.concatenate([truth[:70], np.array([letter, (letter + 1) & 3], np.uint8), truth[70:]]))
    layers += [truth.copy() for _ in range(3)]
    cons, status = hip.poa_banded_emulate([dict(layers=layers)] * 2 + [dict(layers=layers[:9])], variant=VARIANT[0])
    assert [int(s) & 0xFF for s in status] == [1, 1, 1]
    assert np.array_equal(cons[0], _oracle(dict(layers=layers)))
    assert np.array_equal(cons[2], _oracle(dict(layers=layers[:9])))


def test_limits_are_reported():
    """A layer longer than the banded kernels take (896 bases) comes back as status 4 with the backbone as output, as on
    the GPU, and does not disturb the other windows of its wave."""
    rng = np.random.default_rng(23)
    bb = rng.integers(0, 4, size=950, dtype=np.uint8)
    long_w = dict(layers=[bb, bb.copy(), bb.copy()])
    ok_w = _window(rng, 120, 6)
    cons, status = hip.poa_banded_emulate([long_w, ok_w, long_w, ok_w, ok_w], variant=VARIANT[0])
    for i in (0, 2):
        assert (int(status[i]) & 0xFF) == 4 and np.array_equal(cons[i], bb)
    for i in (1, 3, 4):
        assert (int(status[i]) & 0xFF) == 1 and np.array_equal(cons[i], _oracle(ok_w))


@pytest.mark.parametrize("fill", [1, 3, 2], ids=["off_diagonal_scores", "band_edge_scores", "huge_scores"])
def test_what_the_score_ring_holds_beside_a_band_does_not_matter(fill, monkeypatch):
    """A ring row is the 32 cells of its band, addressed by the column (poa4.hip): a read right of a
    predecessor's band, or left of it in a first column, finds another cell of that ring row where rounds 3-4 kept pads of
    -inf.  The claim is that this can only send a window on to the 64-column kernel, never change a consensus: a raised
    candidate that wins is followed by the traceback out of the band and found out.  Here the rings start every layer as
    scores of cells far off the diagonal (-100 .. -900: what such a read finds in practice) / as scores of a band's
    edge in a window's first rows (-40 .. 10: some first columns are raised, some are not — 10 of the 24 windows still
    polish, all of them exactly) / as scores above any real one instead of -inf (RVN_POA4_RING_FILL, emulator build only): whatever
    still comes back polished must equal the oracle, and with off-diagonal leftovers nearly everything still does (with huge
    ones nothing: every first column is raised, every walk is found out)."""
    rng = np.random.default_rng(29)
    wins = [_window(rng, int(rng.integers(90, 330)), int(rng.integers(5, 16)), partial=0.3 if i % 2 else 0.0) for i in range(24)]
    _, base = hip.poa_banded_emulate(wins, variant=VARIANT[0])
    monkeypatch.setenv("RVN_POA4_RING_FILL", str(fill))
    cons, status = hip.poa_banded_emulate(wins, variant=VARIANT[0])
    polished = 0
    for w, c, st, b in zip(wins, cons, status, base):
        if (int(st) & 0xFF) == 1:
            polished += 1
            assert (int(b) & 0xFF) == 1
            assert np.array_equal(c, _oracle(w))
        else:
            assert (int(st) & 0xFF) == 8, st
    base_polished = sum((int(b) & 0xFF) == 1 for b in base)
    if fill == 1:
        assert polished >= base_polished - 4, (polished, base_polished)
    if fill == 3:
        assert 0 < polished < base_polished, (polished, base_polished)  # (the case that tells: both outcomes occur)
    print("ring fill", fill, "polished", polished, "of", base_polished)


def _fan_in_window(rng, length=160, noise=0.0):
    """A window whose backbone position P collects more than eight in-edges: deletions of 1..7 bases in front of P (tails
    P - 2 .. P - 8), every letter and two pairs of letters inserted in front of it (six more tails) — each variant twice, the
    second time in another order, so that later layers walk the in-edges of both groups the kernel keeps them in."""
    P = int(rng.integers(40, length - 40))
    truth = rng.integers(0, 4, size=length, dtype=np.uint8)
    var = [np.concatenate([truth[:P - d], truth[P:]]) for d in range(1, 8)]
    var += [np.concatenate([truth[:P], [c], truth[P:]]).astype(np.uint8) for c in range(4)]
    var += [np.concatenate([truth[:P], [c, 3 - c], truth[P:]]).astype(np.uint8) for c in range(2)]
    layers = [truth.copy()]
    for _ in range(2):
        for i in rng.permutation(len(var)):
            layers.append(_mutate(rng, var[i], noise, noise / 2, noise / 2) if noise else var[i].copy())
        for _ in range(3):
            layers.append(_mutate(rng, truth, noise, noise / 2, noise / 2) if noise else truth.copy())
    return dict(layers=layers)


@pytest.mark.parametrize("fill", [0, 1], ids=["rings_start_low", "off_diagonal_scores"])
def test_rows_of_nine_to_fifteen_in_edges(fill, monkeypatch):
    """A graph row with more than eight in-edges used to send its window to the 64-column kernel (241 of the 244 windows a C4
    round handed on: a second launch behind the persistent one).  Such a row now keeps in-edges 7..14 in an overflow record;
    the NW's rare path folds them as a second group and a per-row mask tells the traceback which group a code counts in
    (poa4.hip, P4::kEdgesMax).  The clean windows here have a row of >= 9 in-edges by construction (the kernel before this
    change flagged them with reason 3) and must be polished, exactly; the noisy ones may still meet another limit."""
    if fill:
        monkeypatch.setenv("RVN_POA4_RING_FILL", str(fill))
    rng = np.random.default_rng(0)
    wins = [_fan_in_window(rng, noise=0.0 if i < 4 else 0.02) for i in range(8)]
    cons, status = hip.poa_banded_emulate(wins, variant=VARIANT[0])
    for i, (w, c, st) in enumerate(zip(wins, cons, status)):
        if (int(st) & 0xFF) == 1:
            assert np.array_equal(c, _oracle(w)), i
        else:
            assert (int(st) & 0xFF) == 8 and (fill or i >= 4) and ((int(st) >> 24) & 15) != 3, (i, hex(int(st)))
    assert sum((int(st) & 0xFF) == 1 for st in status) >= (4 if fill else 7)
