"""GPU polishing round (rvn_polish_round: device mapping + anchors -> windows -> POA kernel -> stitch) against the
CPU restatement of racon's round (whole-overlap NW path -> CIGAR breakpoints).  Tolerance parity (north_star:
'polished consensus within stated edit-distance tolerance'): ED(gpu, cpu) <= 0.2 % of the target + 10 and
ED(gpu, truth) <= 1.1 x ED(cpu, truth) + 10 (measured: 0-1 edits apart with trimming, tools/eval_polish.py)."""
import numpy as np
import pytest

from oracle import oracle
from raven_amd import hip, seqio
from tests import polish_util as pu2

pytestmark = pytest.mark.gpu


def _ed(a, b):
    return oracle.edit_distance(bytes(np.asarray(a, np.uint8) + 65), bytes(np.asarray(b, np.uint8) + 65))


@pytest.mark.parametrize("with_qual,n_targets", [(False, 1), (True, 1), (False, 3)])
def test_polish_round_matches_cpu_within_tolerance(with_qual, n_targets):
    truths, drafts, targets, reads, quals = pu2.make_case(genome_len=24_000, coverage=25, read_len=2500, seed=7,
                                                         with_qual=with_qual, n_targets=n_targets)
    eng = hip.Engine(15, 5)
    td, rd = eng.upload(targets), eng.upload(reads)
    q = 10.0 if with_qual else 0.0
    cons, ratio, st = eng.polish_round(td, rd, quals=quals, q=q)
    ref, ref_ratio = oracle.polish_round(targets, reads, quals=quals, q=q)
    assert st["n_failed_windows"] == 0 and st["n_reads_used"] > 0.8 * reads.n
    for t in range(n_targets):
        assert ratio[t] > 0.85 and abs(ratio[t] - ref_ratio[t]) < 0.1
        ed_draft, ed_cpu, ed_gpu = _ed(drafts[t], truths[t]), _ed(ref[t], truths[t]), _ed(cons[t], truths[t])
        assert ed_gpu < ed_draft, (ed_draft, ed_gpu)
        assert ed_gpu <= 1.1 * ed_cpu + 10, (ed_draft, ed_cpu, ed_gpu)
        assert _ed(cons[t], ref[t]) <= 0.002 * len(ref[t]) + 10, (len(ref[t]), _ed(cons[t], ref[t]))


def test_polish_two_rounds_and_low_quality_reads_are_dropped():
    truths, drafts, targets, reads, quals = pu2.make_case(genome_len=16_000, coverage=20, read_len=2000, seed=9,
                                                         with_qual=True)
    eng = hip.Engine(15, 5)
    rd = eng.upload(reads)
    cur = targets
    eds = [_ed(drafts[0], truths[0])]
    for _ in range(2):  # raven's default num_rounds = 2 (polish.hpp:28)
        cons, ratio, _ = eng.polish_round(eng.upload(cur), rd, quals=quals, q=10.0, trim=False)
        eds.append(_ed(cons[0], truths[0]))
        cur = seqio.pack_reads([cons[0]])
    assert eds[1] < 0.5 * eds[0] and eds[2] <= eds[1] + 10, eds
    # quality threshold above every read's mean quality (12): no layer survives -> nothing is polished (racon)
    cons, ratio, st = eng.polish_round(eng.upload(targets), rd, quals=quals, q=20.0)
    assert st["n_layers"] == 0 and ratio[0] == 0.0 and np.array_equal(cons[0], drafts[0])


@pytest.mark.parametrize("with_qual", [True, False])
def test_lambda_polish_matches_golden_fixture(with_qual):
    """The reference's own test data (RavenTest/data: ERA476754 reads, NC_001416): one polishing round of a seeded
    draft of lambda on the device vs the committed oracle result (tests/golden/lambda_polish.npz), with Raven's
    block-mean qualities and its avg_q threshold (polish.cc:25-47) and without qualities."""
    import importlib.util
    import os
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_polish", os.path.join(golden, "make_golden_polish.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    rs, truth, draft, quals, avg_q = mg.inputs()
    fx = np.load(os.path.join(golden, "lambda_polish.npz"))
    eng = hip.Engine(15, 5)
    cons, ratio, st = eng.polish_round(eng.upload(seqio.pack_reads([draft])), eng.upload(rs),
                                       quals=quals if with_qual else None, q=avg_q if with_qual else 0.0)
    if not with_qual:
        # real reads without the quality filter need more than the 128-column band for some windows (256 columns or the
        # full-matrix kernel; with small chunks the latter are collected and run in one final batch): no byte may change
        assert eng.poa_fallback_windows() >= 1
        eng.polish_set_chunk_windows(16)
        cons2, ratio2, _ = eng.polish_round(eng.upload(seqio.pack_reads([draft])), eng.upload(rs))
        assert np.array_equal(cons2[0], cons[0]) and ratio2[0] == ratio[0]
    ref = fx["consensus" if with_qual else "consensus_noqual"]
    ed_ref = int(fx["ed_consensus" if with_qual else "ed_consensus_noqual"][0])
    assert st["n_failed_windows"] == 0 and ratio[0] == 1.0
    d = _ed(cons[0], ref)
    assert d <= 0.0025 * len(ref) + 10, (d, len(ref), len(cons[0]))  # measured 35 (qual) / 77-90 (no qual) of 47.8 kb
    assert _ed(cons[0], truth) <= 1.1 * ed_ref + 10


def test_chunked_pipeline_gives_the_same_round():
    """The round is processed in window chunks (host cuts of chunk i+1 overlap the POA of chunk i): the chunk size
    must not change a byte, for several targets and with the quality filter."""
    truths, drafts, targets, reads, quals = pu2.make_case(genome_len=30_000, coverage=20, read_len=2500, seed=13,
                                                         with_qual=True, n_targets=2)
    eng = hip.Engine(15, 5)
    td, rd = eng.upload(targets), eng.upload(reads)
    assert eng.polish_set_chunk_windows(0) == 16384
    ref, ref_ratio, st0 = eng.polish_round(td, rd, quals=quals, q=10.0)
    for chunk in (1, 7, 32):
        eng.polish_set_chunk_windows(chunk)
        cons, ratio, st = eng.polish_round(td, rd, quals=quals, q=10.0)
        assert np.allclose(ratio, ref_ratio) and st["n_layers"] == st0["n_layers"] and st["n_windows"] == st0["n_windows"]
        assert st["n_reads_used"] == st0["n_reads_used"]
        for a, b in zip(cons, ref):
            assert np.array_equal(a, b)
