"""The second mapping pass (rvn_find_overlaps_and_repetitive_regions = raven::FindOverlapsAndRepetetiveRegions,
RavenLib/src/construct.cc:316-491) and the identity filter loop of ResolveContainedReads (construct.cc:162-217) on the
device, bit-exact against the oracle's restatement: the overlap list the reference leaves in overlaps.back(), the piles
marked as contained and every pile's k-mer cells (Pile::AddKmers)."""
import numpy as np
import pytest

from oracle import oracle
from raven_amd import hip, seqio, synth

pytestmark = pytest.mark.gpu


def _repeat_genome(n, seed):
    """Genome with a 1.5 kb repeat in 6 copies at 99 % identity: exercises Filter, `filtered` and AddKmers."""
    rng = np.random.default_rng(seed)
    g = synth.make_genome(n, seed=seed)
    rep = g[1000:2500].copy()
    for c in range(1, 6):
        at = 1000 + c * (n // 6)
        g[at:at + 1500] = synth.mutate(rng, rep, 0.01, 0.0, 0.0)[:1500]
    return g


def _pass1_regions(eng, rd):
    """Valid regions as raven computes them after the first pass (TrimAndAnnotatePiles: FindValidRegion(4))."""
    p = eng.find_overlaps_and_create_piles(rd)
    begin, end, median, invalid = p.trim_and_annotate(4)
    p.close()
    return (begin.astype(np.uint32) << 4), (end.astype(np.uint32) << 4), invalid.astype(np.uint8)


def _compare(got, want, rs):
    assert np.array_equal(got["contained"], want["contained"])
    assert got["overlaps"].shape == want["overlaps"].shape
    assert np.array_equal(got["overlaps"], want["overlaps"].astype(hip.OVERLAP_DTYPE))
    for i in range(rs.n):
        assert np.array_equal(got["kmers"][i], want["kmers"][i]), i


@pytest.mark.parametrize("identity,batch_bases", [(0.0, 1 << 30), (0.0, 150_000), (0.78, 1 << 30), (0.78, 120_000)])
def test_second_pass_matches_oracle(identity, batch_bases):
    g = _repeat_genome(60_000, seed=3)
    rs, _ = synth.make_reads(g, 14, 2500, seed=4)
    eng = hip.Engine(15, 5)
    rd = eng.upload(rs)
    begin, end, invalid = _pass1_regions(eng, rd)
    rng = np.random.default_rng(8)
    invalid = (invalid | (rng.random(rs.n) < 0.15)).astype(np.uint8)  # some piles already invalid (contained reads)
    got = eng.find_overlaps_and_repetitive_regions(rd, begin, end, invalid, freq=0.01, identity=identity,
                                                   batch_bases=batch_bases)
    want = oracle.second_pass(15, 5, rs, begin, end, invalid, freq=0.01, identity=identity, batch_bases=batch_bases)
    _compare(got, want, rs)
    assert got["overlaps"].shape[0] > 50 and got["contained"].sum() > 0
    assert sum(int(k.sum()) for k in got["kmers"]) > 0  # repeats produced filtered minimizers
    if identity:
        loose = eng.find_overlaps_and_repetitive_regions(rd, begin, end, invalid, freq=0.01, identity=0.0,
                                                         batch_bases=batch_bases)
        assert loose["overlaps"].shape[0] > got["overlaps"].shape[0]  # the filter dropped something


def test_second_pass_degenerate_inputs():
    g = synth.make_genome(30_000, seed=9)
    rs, _ = synth.make_reads(g, 10, 2000, seed=10)
    eng = hip.Engine(15, 5)
    rd = eng.upload(rs)
    full_b = np.zeros(rs.n, np.uint32)
    full_e = ((rs.lengths >> 4) << 4).astype(np.uint32)
    # every pile invalid: nothing to do
    got = eng.find_overlaps_and_repetitive_regions(rd, full_b, full_e, np.ones(rs.n, np.uint8))
    assert got["overlaps"].shape[0] == 0 and got["contained"].sum() == 0 and all(len(k) == 0 for k in got["kmers"])
    # untrimmed piles, everything valid: the reference's `s` (construct.cc:343-349, position of the first invalid pile)
    # stays 0, so it maps NOTHING — reproduced, not "fixed"
    inv = np.zeros(rs.n, np.uint8)
    got = eng.find_overlaps_and_repetitive_regions(rd, full_b, full_e, inv)
    want = oracle.second_pass(15, 5, rs, full_b, full_e, inv)
    assert want["overlaps"].shape[0] == 0
    _compare(got, want, rs)
    # one invalid pile is enough for the pass to run over all the valid ones
    inv[rs.n // 2] = 1
    got = eng.find_overlaps_and_repetitive_regions(rd, full_b, full_e, inv)
    want = oracle.second_pass(15, 5, rs, full_b, full_e, inv)
    assert want["overlaps"].shape[0] > 0
    _compare(got, want, rs)


@pytest.mark.parametrize("identity", [0.7, 0.8, 0.9])
def test_identity_filter_of_contained_read_resolution_matches_oracle(identity):
    g = synth.make_genome(50_000, seed=21)
    rs, _ = synth.make_reads(g, 12, 2500, seed=22)
    eng = hip.Engine(15, 5)
    rd = eng.upload(rs)
    p = eng.find_overlaps_and_create_piles(rd)
    ovl, off = p.overlaps()
    begin, end, median, invalid = p.trim_and_annotate(4)
    p.close()
    begin, end = (begin.astype(np.uint32) << 4), (end.astype(np.uint32) << 4)
    got_o, got_off = eng.filter_overlaps_by_identity(rd, ovl, off, begin, end, invalid, identity)
    want_o, want_off = oracle.identity_filter(rs, ovl.astype(oracle.OVERLAP_DTYPE), off, begin, end, invalid, identity)
    assert np.array_equal(got_off, want_off)
    assert np.array_equal(got_o, want_o.astype(hip.OVERLAP_DTYPE))
    assert got_o.shape[0] <= ovl.shape[0]
    if identity < 0.85:  # ONT-like reads are ~80 % identical to each other: nothing passes 0.9
        assert got_o.shape[0] > 0
