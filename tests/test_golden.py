"""Committed golden vectors (tests/golden/lambda_pass1.npz, made by tests/golden/make_golden.py from the
reference's own lambda reads): the oracle must still reproduce them (CPU), and the HIP path must match them
on the GPU (through the C ABI)."""
import os

import numpy as np
import pytest

from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = os.path.join(GOLDEN, "lambda_pass1.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_fixture_inputs(lambda_reads, lambda_genome, gold):
    # the data files the reference's own tests hold (raven_test.cpp:21-35): 236 reads / 1,674,628 bases; 48,502 bp
    assert lambda_reads.n == 236 == int(gold["n_reads"][0])
    assert lambda_reads.total_bases == 1674628 == int(gold["total_bases"][0])
    assert lambda_genome.n == 1 and int(lambda_genome.lengths[0]) == 48502
    assert int(lambda_reads.lengths.min()) == 443 and int(lambda_reads.lengths.max()) == 11968


def test_oracle_reproduces_golden(lambda_reads, gold):
    e = oracle.Engine(15, 5)
    for i in range(4):
        for mh in (0, 1):
            v, o = e.sketch(lambda_reads, i, bool(mh))
            assert np.array_equal(v, gold["sketch_%d_%d_values" % (i, mh)])
            assert np.array_equal(o, gold["sketch_%d_%d_origins" % (i, mh)])
    for mh in (0, 1):
        r = oracle.Engine(15, 5).find_overlaps_and_create_piles(lambda_reads, use_minhash=bool(mh))
        assert r["occurrence"] == int(gold["pass1_%d_occurrence" % mh][0])
        assert np.array_equal(r["overlaps"], gold["pass1_%d_overlaps" % mh])
        assert np.array_equal(r["overlap_offsets"], gold["pass1_%d_overlap_offsets" % mh])
        assert np.array_equal(r["pile_data"], gold["pass1_%d_pile_data" % mh])


@pytest.mark.gpu
def test_hip_matches_golden(lambda_reads, gold):
    from raven_amd import hip
    eng = hip.Engine(15, 5)
    rd = eng.upload(lambda_reads)
    for mh in (0, 1):
        v, o, off = eng.sketch(rd, 0, 4, bool(mh))
        for i in range(4):
            assert np.array_equal(v[off[i]:off[i + 1]], gold["sketch_%d_%d_values" % (i, mh)])
            assert np.array_equal(o[off[i]:off[i + 1]], gold["sketch_%d_%d_origins" % (i, mh)])
    for mh in (0, 1):
        p = eng.find_overlaps_and_create_piles(rd, use_minhash=bool(mh))
        ovl, off = p.overlaps()
        data, poff = p.piles()
        assert eng.occurrence == int(gold["pass1_%d_occurrence" % mh][0])
        assert np.array_equal(ovl, gold["pass1_%d_overlaps" % mh])
        assert np.array_equal(off.astype(np.uint64), gold["pass1_%d_overlap_offsets" % mh])
        assert np.array_equal(data, gold["pass1_%d_pile_data" % mh])
        assert np.array_equal(poff, gold["pass1_%d_pile_offsets" % mh])
        p.close()


def test_lambda_polish_fixture_is_reproducible_and_improves_the_draft():
    """tests/golden/lambda_polish.npz (oracle polishing round on the reference's own lambda data): inputs rebuild
    byte-identically from the committed generator, and the stored distances are what the consensus really has."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_polish", os.path.join(GOLDEN, "make_golden_polish.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    rs, truth, draft, quals, avg_q = mg.inputs()
    fx = np.load(os.path.join(GOLDEN, "lambda_polish.npz"))
    assert np.array_equal(draft, fx["draft"]) and abs(avg_q - float(fx["avg_q"][0])) < 1e-9
    assert 9.0 < avg_q < 12.0 and all(len(q) == int(n) for q, n in zip(quals, rs.lengths))
    t = bytes(truth + 65)
    assert oracle.edit_distance(bytes(fx["consensus"] + 65), t) == int(fx["ed_consensus"][0])
    assert int(fx["ed_consensus"][0]) < int(fx["ed_consensus_noqual"][0]) < int(fx["ed_draft"][0])
    assert float(fx["ratio"][0]) == 1.0
