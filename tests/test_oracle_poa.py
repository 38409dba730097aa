"""CPU pins for the POA window-consensus oracle (oracle/poa_oracle.cpp = racon Window::GenerateConsensus over a
spoa-style graph): definitional properties, since racon/spoa themselves are not available (parity unpinned)."""
import numpy as np
import pytest

from oracle import oracle


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


def _nw_score(t, q, m=3, n=-5, g=-4):
    H = np.zeros((len(t) + 1, len(q) + 1), dtype=np.int64)
    H[0, :] = np.arange(len(q) + 1) * g
    H[:, 0] = np.arange(len(t) + 1) * g
    for i in range(1, len(t) + 1):
        for j in range(1, len(q) + 1):
            H[i, j] = max(H[i - 1, j - 1] + (m if t[i - 1] == q[j - 1] else n), H[i - 1, j] + g, H[i, j - 1] + g)
    return int(H[-1, -1])


def _ed(a, b):
    return oracle.edit_distance(bytes(np.asarray(a, np.uint8) + 65), bytes(np.asarray(b, np.uint8) + 65))


def test_align_score_on_linear_graph_matches_plain_nw():
    rng = np.random.default_rng(0)
    for _ in range(20):
        t = rng.integers(0, 4, size=int(rng.integers(1, 60)), dtype=np.uint8)
        q = _mutate(rng, t, 0.1, 0.1, 0.1) if rng.random() < 0.7 else rng.integers(0, 4, size=int(rng.integers(1, 60)), dtype=np.uint8)
        if len(q) == 0:
            continue
        assert oracle.poa_align_score_linear(t, q) == _nw_score(t, q)


def test_fewer_than_three_sequences_returns_backbone():
    bb = np.array([0, 1, 2, 3, 0, 1], np.uint8)
    cons, polished = oracle.poa_window([bb, np.array([0, 1, 2, 3, 3, 1], np.uint8)])
    assert not polished and np.array_equal(cons, bb)
    cons, polished = oracle.poa_window([bb])
    assert not polished and np.array_equal(cons, bb)


def test_error_free_layers_fix_a_noisy_backbone():
    rng = np.random.default_rng(1)
    truth = rng.integers(0, 4, size=500, dtype=np.uint8)
    bb = _mutate(rng, truth, 0.03, 0.02, 0.02)
    layers = [bb] + [truth.copy() for _ in range(6)]
    cons, polished = oracle.poa_window(layers, begins=[0] * 7, ends=[len(bb) - 1] * 7)
    assert polished and np.array_equal(cons, truth)


def test_majority_of_noisy_layers_beats_each_layer():
    rng = np.random.default_rng(2)
    truth = rng.integers(0, 4, size=500, dtype=np.uint8)
    bb = _mutate(rng, truth, 0.04, 0.03, 0.03)
    reads = [_mutate(rng, truth, 0.04, 0.03, 0.03) for _ in range(30)]
    cons, polished = oracle.poa_window([bb] + reads, begins=[0] * 31, ends=[len(bb) - 1] * 31)
    assert polished
    ed_cons = _ed(cons, truth)
    assert ed_cons <= 0.2 * min(_ed(r, truth) for r in reads + [bb])
    assert ed_cons <= 10


def test_quality_weights_decide_ties():
    # 2 layers say 'A' with high quality, 2 say 'C' with low quality at one column: heavy path follows quality
    truth = np.tile(np.array([0, 1, 2, 3], np.uint8), 10)
    bad = truth.copy()
    bad[20] = (bad[20] + 1) & 3
    layers = [truth, truth, truth, bad, bad]
    hi = np.full(40, 33 + 40, np.uint8)
    lo = np.full(40, 33 + 2, np.uint8)
    cons, _ = oracle.poa_window(layers, quals=[np.full(40, 33, np.uint8), hi, hi, lo, lo], trim=False)
    assert np.array_equal(cons, truth)
    cons, _ = oracle.poa_window([bad, bad, bad, truth, truth], quals=[np.full(40, 33, np.uint8), lo, lo, hi, hi], trim=False)
    assert np.array_equal(cons, truth)


def test_partial_layers_use_subgraph_and_trim():
    rng = np.random.default_rng(3)
    truth = rng.integers(0, 4, size=500, dtype=np.uint8)
    bb = truth.copy()
    bb[100] = (bb[100] + 1) & 3
    bb[400] = (bb[400] + 2) & 3
    layers, begins, ends = [bb], [0], [499]
    for b, e in ((0, 250), (0, 260), (30, 300), (200, 500), (220, 500), (240, 499), (0, 500), (0, 500)):
        layers.append(truth[b:e].copy())
        begins.append(b)
        ends.append(e - 1 if e < 500 else 499)  # racon passes the last covered backbone position
    cons, polished = oracle.poa_window(layers, begins=begins, ends=ends, trim=True)
    assert polished and np.array_equal(cons, truth)
    # with trimming off a thin layer set keeps the full backbone span
    cons2, _ = oracle.poa_window(layers, begins=begins, ends=ends, trim=False)
    assert np.array_equal(cons2, truth)


def test_trim_cuts_low_coverage_ends():
    rng = np.random.default_rng(4)
    truth = rng.integers(0, 4, size=300, dtype=np.uint8)
    layers, begins, ends = [truth.copy()], [0], [299]
    for _ in range(8):  # every read covers only the middle
        layers.append(truth[50:250].copy())
        begins.append(50)
        ends.append(249)
    cons, _ = oracle.poa_window(layers, begins=begins, ends=ends, trim=True)
    assert np.array_equal(cons, truth[50:250])
    cons, _ = oracle.poa_window(layers, begins=begins, ends=ends, trim=False)
    assert np.array_equal(cons, truth)


def test_incremental_topological_order_rule_stays_valid():
    """The device kernel keeps the topological order incrementally (new nodes go after the whole aligned group of
    their anchor column).  Replayed here next to spoa's graph construction: no edge may ever point backwards."""
    rng = np.random.default_rng(11)
    for _ in range(40):
        truth = rng.integers(0, 4, size=int(rng.integers(200, 500)), dtype=np.uint8)
        bb = _mutate(rng, truth, 0.03, 0.02, 0.02)
        layers, begins, ends = [bb], [0], [len(bb) - 1]
        for _ in range(int(rng.integers(10, 32))):
            if rng.random() < 0.25:
                b0 = int(rng.integers(0, len(truth) // 2))
                e0 = int(rng.integers(b0 + len(truth) // 4, len(truth)))
            else:
                b0, e0 = 0, len(truth)
            layers.append(_mutate(rng, truth[b0:e0], 0.05, 0.04, 0.04))
            bb_b = min(len(bb) - 2, int(b0 * len(bb) / len(truth)))
            begins.append(bb_b)
            ends.append(min(len(bb) - 1, max(bb_b + 1, int(e0 * len(bb) / len(truth)) - 1)))
        bad, info = oracle.poa_order_check(layers, begins, ends)
        assert bad == -1, (bad, info)


def test_polish_round_oracle_improves_draft():
    """racon round restatement: polishing a 1-2 % error draft with 25x ONT-like reads removes most errors."""
    from tests import polish_util as pu2
    truths, drafts, targets, reads, _ = pu2.make_case(genome_len=20_000, coverage=25, read_len=2500, seed=5)
    cons, ratio = oracle.polish_round(targets, reads)
    assert ratio[0] > 0.9
    ed_draft = _ed(drafts[0], truths[0])
    ed_pol = _ed(cons[0], truths[0])
    # racon's TGS trimming cuts the low-coverage contig ends: errors = edits beyond the plain length loss
    lost = len(truths[0]) - len(cons[0])
    assert ed_draft > 300 and 0 <= lost < 800 and ed_pol - lost < 0.2 * ed_draft, (ed_draft, ed_pol, lost)
    untrimmed, _ = oracle.polish_round(targets, reads, trim=False)
    assert _ed(untrimmed[0], truths[0]) < 0.2 * ed_draft
    # second round on the polished sequence does not make it worse
    from raven_amd import seqio
    cons2, _ = oracle.polish_round(seqio.pack_reads([cons[0]]), reads)
    assert _ed(cons2[0], truths[0]) - (len(truths[0]) - len(cons2[0])) <= ed_pol - lost + 25
