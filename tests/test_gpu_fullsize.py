"""BASELINE.json's configs at FULL size on one MI355X, through size-independent properties (the oracle cannot run at
these sizes): configs[2] (5 Mb, 30x ONT 10 kb, -p 2), the configs[3] workload (100 Mb, 30x ONT-length reads, -p 2; on
one GPU — three flush windows) and the configs[4] workload (100 Mb, 40x HiFi 15 kb, --identity 0.95, -p 2).
Data comes from the seeded generator on the GPU (raven_amd/synth.py, torch as the random engine), ground truth included."""
import math

import numpy as np
import pytest

from oracle import oracle
from raven_amd import hip, synth

pytestmark = pytest.mark.gpu


def _ed(a, b):
    return oracle.edit_distance(bytes(np.asarray(a, np.uint8) + 65), bytes(np.asarray(b, np.uint8) + 65))


def _make(genome_bases, coverage, read_len, model, errs, seed):
    import torch
    dev = torch.device("cuda", 0)
    g = synth.make_genome_torch(genome_bases, seed=seed, device=dev)
    rs, truth = synth.make_reads_torch(g, coverage, read_len, length_model=model, sub=errs[0], ins=errs[1], dele=errs[2],
                                       seed=seed + 1)
    return g, rs, truth


def _drafts(g, contig, seed):
    n = int(g.shape[0])
    bounds = np.linspace(0, n, max(1, n // contig) + 1).astype(np.int64)
    out = []
    for i in range(len(bounds) - 1):
        out.append(synth.mutate_torch(g[int(bounds[i]):int(bounds[i + 1])], 0.01, 0.008, 0.008, seed=seed + i).cpu().numpy())
    return out, bounds


def _pass1_properties(eng, rd, rs, truth, kmax=32, min_precision=0.999):
    p = eng.find_overlaps_and_create_piles(rd, kmax=kmax)
    ovl, off = p.overlaps()
    data, poff = p.piles()
    cnt = np.diff(off.astype(np.int64))
    # structure: CSR consistent, <= kmax per pile, ids consistent, spans inside the reads
    assert off[0] == 0 and off[-1] == ovl.shape[0] and cnt.max() <= kmax
    pile_of = np.repeat(np.arange(rs.n, dtype=np.uint32), cnt)
    assert np.array_equal(ovl["lhs_id"], pile_of) and np.all(ovl["rhs_id"] != ovl["lhs_id"])
    assert np.all(ovl["lhs_end"] <= rs.lengths[ovl["lhs_id"]]) and np.all(ovl["rhs_end"] <= rs.lengths[ovl["rhs_id"]])
    assert np.all(ovl["score"] >= 100)
    # truncated lists are sorted by decreasing overlap length
    lens = np.maximum(ovl["lhs_end"] - ovl["lhs_begin"], ovl["rhs_end"] - ovl["rhs_begin"]).astype(np.int64)
    full = np.nonzero(cnt == kmax)[0]
    assert full.size > rs.n // 4
    for pidx in full[:: max(1, full.size // 1500)]:
        assert np.all(np.diff(lens[off[pidx]: off[pidx + 1]]) <= 0)
    # ground truth: reported overlaps are real overlaps of the source segments, on the right strand
    s, e_ = truth["start"], truth["start"] + truth["src_len"]
    inter = np.minimum(e_[ovl["lhs_id"]], e_[ovl["rhs_id"]]) - np.maximum(s[ovl["lhs_id"]], s[ovl["rhs_id"]])
    assert (inter > 0).mean() >= min_precision
    same = truth["strand"][ovl["lhs_id"]] == truth["strand"][ovl["rhs_id"]]
    assert (same == (ovl["strand"] == 1))[inter > 500].mean() > 0.999
    assert data.max() < 65535
    # idempotence: a second pass gives byte-identical results
    p2 = eng.find_overlaps_and_create_piles(rd, kmax=kmax)
    ovl2, off2 = p2.overlaps()
    data2, _ = p2.piles()
    assert np.array_equal(ovl, ovl2) and np.array_equal(off, off2) and np.array_equal(data, data2)
    p2.close()
    return p, ovl, off, data


def _polish_properties(eng, rd, g, drafts, bounds, rounds, n_slices=3):
    """`rounds` racon rounds; ED to the truth on a few 20 kb slices must shrink round over round."""
    truth = g.cpu().numpy()
    cur = drafts
    slices = [int(x) for x in np.linspace(0, len(drafts) - 1, n_slices).astype(int)]

    def errors(seq, c):
        # the first 20 kb of contig c against a truth prefix that is 200 bases longer (the draft's indels shift the end):
        # the distance is 200 + the errors of the prefix
        return _ed(seq[:20_000], truth[int(bounds[c]):int(bounds[c]) + 20_200]) - 200

    eds = [[errors(cur[c], c) for c in slices]]
    stats = []
    for _ in range(rounds):
        td = eng.upload_codes(cur)
        cons, ratio, st = eng.polish_round(td, rd)
        td.close()
        assert st["n_failed_windows"] == 0 and st["n_dropped_layers"] == 0
        assert st["n_polished_windows"] >= 0.999 * st["n_windows"] and min(ratio) > 0.99
        assert st["n_aligned"] == st["n_reads_used"] > 0.95 * rd.n
        cur = cons
        eds.append([errors(cur[c], c) for c in slices])
        stats.append(st)
    return cur, eds, stats


def test_configs2_polish_full_size():
    """configs[2]: 5 Mb, 30x ONT 10 kb reads, -p 2 -> 10 000 windows per round."""
    g, rs, truth = _make(5_000_000, 30, 10000, "fixed", (0.04, 0.03, 0.03), 0x5EED0001)
    eng = hip.Engine(15, 5)
    rd = eng.upload(rs)
    drafts, bounds = _drafts(g, 5_000_000, 77)
    cons, eds, stats = _polish_properties(eng, rd, g, drafts, bounds, rounds=2, n_slices=1)
    assert stats[0]["n_windows"] >= 10_000
    # the prefix gets closer to the truth every round
    assert eds[1][0] < 0.4 * eds[0][0] and eds[2][0] <= eds[1][0] + 5, eds
    # window-range invariance: the round in two halves (what two GPUs would do) reproduces it byte for byte
    td = eng.upload_codes(drafts)
    whole, _, st = eng.polish_round(td, rd)
    nw = st["n_windows"]
    a, _, _, _ = eng.polish_round_range(td, rd, 0, nw // 2)
    b, _, _, _ = eng.polish_round_range(td, rd, nw // 2, nw)
    assert np.array_equal(np.concatenate([a[0], b[0]]), whole[0])


def test_configs3_workload_on_one_gpu():
    """configs[3]'s workload (100 Mb, 30x ONT-length reads, -p 2) on one GPU: 3 query flush windows of 2^30 bases."""
    g, rs, truth = _make(100_000_000, 30, 9000, "lognormal", (0.04, 0.03, 0.03), 0x5EED0011)
    assert rs.total_bases > 2 * (1 << 30)  # three flush windows (construct.cc:66-70)
    eng = hip.Engine(15, 5)
    rd = eng.upload(rs)
    p, ovl, off, data = _pass1_properties(eng, rd, rs, truth)
    assert 12 < data.mean() < 45
    p.close()
    drafts, bounds = _drafts(g, 5_000_000, 99)
    cons, eds, stats = _polish_properties(eng, rd, g, drafts, bounds, rounds=2)  # -p 2
    assert stats[0]["n_windows"] >= 200_000 and stats[1]["n_windows"] >= 199_000
    for before, after, second in zip(eds[0], eds[1], eds[2]):
        assert after < 0.4 * before and second <= after + 5, eds


def test_configs4_workload_hifi_identity_on_one_gpu():
    """configs[4]'s workload (100 Mb, 40x HiFi 15 kb reads, --identity 0.95, -p 2) on one GPU: first pass, the
    identity filter of ResolveContainedReads on its overlap lists, one polishing round."""
    g, rs, truth = _make(100_000_000, 40, 15000, "normal", (0.001, 0.002, 0.002), 0x5EED0021)
    eng = hip.Engine(15, 5)
    rd = eng.upload(rs)
    p, ovl, off, data = _pass1_properties(eng, rd, rs, truth)
    begin, end, median, invalid = p.trim_and_annotate(4)
    p.close()
    assert invalid.mean() < 0.02 and 20 < np.median(median[~invalid]) < 60
    begin, end = (begin.astype(np.uint32) << 4), (end.astype(np.uint32) << 4)
    kept, koff = eng.filter_overlaps_by_identity(rd, ovl, off, begin, end, invalid, 0.95)
    # HiFi reads of the same locus are ~99 % identical: the filter keeps (nearly) every overlap that survives
    # OverlapUpdate, and everything it keeps lies inside the valid regions
    assert 0.9 * ovl.shape[0] < kept.shape[0] <= ovl.shape[0]
    assert np.all(kept["lhs_begin"] >= begin[kept["lhs_id"]]) and np.all(kept["lhs_end"] <= end[kept["lhs_id"]])
    assert np.all(kept["rhs_begin"] >= begin[kept["rhs_id"]]) and np.all(kept["rhs_end"] <= end[kept["rhs_id"]])
    # ... and a stricter threshold than the data's identity drops almost everything
    none, _ = eng.filter_overlaps_by_identity(rd, ovl[: int(off[2000])], np.minimum(off, off[2000]), begin, end, invalid, 0.9995)
    assert none.shape[0] < 0.2 * int(off[2000])
    # ---- the WHOLE second pass (FindOverlapsAndRepetetiveRegions, construct.cc:316-491) at this size, --identity 0.95 ----
    # contained reads first, as ResolveContainedReads marks them (types of the first pass's overlaps against the trimmed
    # regions; the oracle's rule function is vectorised C++, the lists themselves come from the device)
    upd, ok, ty = oracle.overlap_update_and_type(ovl.astype(oracle.OVERLAP_DTYPE), begin, end, invalid.astype(np.uint8))
    contained = np.zeros(rs.n, bool)
    contained[upd["lhs_id"][(ok == 1) & (ty == 1)]] = True
    contained[upd["rhs_id"][(ok == 1) & (ty == 2)]] = True
    inv2 = (invalid | contained).astype(np.uint8)
    assert 0.2 < inv2.mean() < 0.98  # 40x of 15 kb reads: most reads are contained in a neighbour
    res = eng.find_overlaps_and_repetitive_regions(rd, begin, end, inv2, freq=0.001, kmer_len=15, identity=0.95)
    o2 = res["overlaps"]
    newly = res["contained"].astype(bool)
    final_invalid = inv2.astype(bool) | newly
    assert o2.shape[0] > 1000
    # every kept overlap joins two piles that are valid at the end, lies inside both valid regions, is long enough ...
    assert not final_invalid[o2["lhs_id"]].any() and not final_invalid[o2["rhs_id"]].any()
    assert np.all(o2["lhs_begin"] >= begin[o2["lhs_id"]]) and np.all(o2["lhs_end"] <= end[o2["lhs_id"]])
    assert np.all(o2["rhs_begin"] >= begin[o2["rhs_id"]]) and np.all(o2["rhs_end"] <= end[o2["rhs_id"]])
    assert np.all(o2["lhs_end"] - o2["lhs_begin"] >= 84) and np.all(o2["rhs_end"] - o2["rhs_begin"] >= 84)
    # ... is a fixed point of OverlapUpdate and a dovetail (GetOverlapType 3 or 4) ...
    upd2, ok2, ty2 = oracle.overlap_update_and_type(o2.astype(oracle.OVERLAP_DTYPE), begin, end, final_invalid.astype(np.uint8))
    assert ok2.all() and np.array_equal(upd2, o2.astype(oracle.OVERLAP_DTYPE)) and np.isin(ty2, (3, 4)).all()
    # ... the consecutive-pair de-duplication holds (construct.cc:440-449), reads marked contained here were valid before
    same_pair = (o2["lhs_id"][1:] == o2["lhs_id"][:-1]) & (o2["rhs_id"][1:] == o2["rhs_id"][:-1])
    assert not same_pair.any()
    assert not (newly & inv2.astype(bool)).any() and newly.sum() > 0
    # ... Pile::kmers_ exists exactly for the piles that entered the pass valid, (len >> 4) + 1 cells each
    sample = np.linspace(0, rs.n - 1, 4000).astype(int)
    for i in sample:
        assert res["kmers"][i].shape[0] == (0 if inv2[i] else (int(rs.lengths[i]) >> 4) + 1), i
    del res, o2, upd, upd2
    drafts, bounds = _drafts(g, 5_000_000, 199)
    cons, eds, stats = _polish_properties(eng, rd, g, drafts, bounds, rounds=2)  # -p 2
    for before, after, second in zip(eds[0], eds[1], eds[2]):
        assert after < 0.1 * before and second <= after + 3, eds  # HiFi layers: (nearly) every draft error goes in one round


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_pass_at_20mb_is_bit_identical_to_the_single_gpu_pass(world):
    """The device-resident sharded pass (3 all-to-all exchanges per flush window, raven_amd/sharded.py) on a 20 Mb genome
    at 30x (600 Mbase, ~67 000 reads, several flush windows of 2^28 bases): every virtual rank's slice of pile coverage
    and truncated overlap lists equals the single-GPU pass bit for bit."""
    import torch
    from raven_amd import sharded
    from tests import sharded_util
    dev = torch.device("cuda", 0)
    g, rs, truth = _make(20_000_000, 30, 9000, "lognormal", (0.04, 0.03, 0.03), 0x5EED0031)
    del g
    eng = hip.Engine(15, 5)
    p = eng.find_overlaps_and_create_piles(eng.upload(rs), flush_bases=1 << 28)
    data, poff = p.piles()
    kept, koff = p.overlaps()
    occ = eng.occurrence
    p.close()
    assert kept.shape[0] > 1_000_000

    def rank_fn(r, comm):
        return sharded.find_overlaps_and_create_piles_sharded_dev(hip.Engine(15, 5), rs, comm, dev, flush_bases=1 << 28)

    res = sharded_util.run_ranks(world, rank_fn)
    assert [x["lo"] for x in res] + [res[-1]["hi"]] == sharded.partition_reads(rs.lengths, world).tolist()
    for x in res:
        assert x["occurrence"] == occ
        sharded_util.check_against_single(x, data, poff, kept, koff)
    assert sum(x["stats"]["matches_sent"] for x in res) > 0 and sum(x["stats"]["overlaps_sent"] for x in res) > 0
