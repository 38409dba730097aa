"""SURVEY 8(a) rows a1 / a18 and 8(f) rank 4 as ONE integration program (tests/cpp/construct_stage_test.cpp): the
reference's stage order through the facades —
  raven::ConstructGraph (RavenLib/src/construct.cc:650-707): FindOverlapsAndCreatePiles -> TrimAndAnnotatePiles ->
      ResolveContainedReads (identity filter via edlibAlign) -> ResolveChimericSequences -> FindOverlapsAndRepetetiveRegions,
  GetUnitigs' names (common.cc:227-252) through racon::Polisher over the rounds of raven::Polish (polish.cc:50-74),
  SalvagePlasmids' Minimize / Filter / Map sequence (assemble.cc:732-795)
— against the same sequences stated with the oracle's primitives here in Python (stage-to-stage hand-off of pile regions,
validity, chimeric regions and overlap lists is what these check; the pieces have their own tests)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle
from raven_amd import hip, seqio, synth
from tests.test_gpu_facade import _build, _write_reads

pytestmark = pytest.mark.gpu


def _clamp(v):
    return int(v) if v < 65535.0 else 65535


def _clear_chimeric_regions(data, begin, end, regions, median):
    """Pile::ClearChimericRegions + UpdateValidRegion (pile.cc:143-156, :189-225) on one pile; cells."""
    best_b = best_e = 0
    last = begin
    unresolved = []
    for (rb, re) in regions:
        if begin > rb or end < re:
            continue
        if any(_clamp(float(data[i]) * 1.82) <= median for i in range(rb, re + 1)):
            if rb - last > best_e - best_b:
                best_b, best_e = last, rb
            last = re
        else:
            unresolved.append((rb, re))
    if end - last > best_e - best_b:
        best_b, best_e = last, end
    chimeric = best_b != begin or best_e != end
    invalid = False
    if best_b >= best_e or best_e - best_b < (1260 >> 4):
        invalid = True
    else:
        data[begin:best_b] = 0
        data[best_e:end] = 0
        begin, end = best_b, best_e
    return begin, end, invalid, chimeric, unresolved


def _oracle_stages(rs, identity):
    n = rs.n
    p1 = oracle.Engine(15, 5).find_overlaps_and_create_piles(rs, freq=0.001, kmax=32, use_minhash=False)
    poff, ooff = p1["pile_offsets"], p1["overlap_offsets"]
    data = [p1["pile_data"][int(poff[i]):int(poff[i + 1])].copy() for i in range(n)]
    lists = [p1["overlaps"][int(ooff[i]):int(ooff[i + 1])].copy() for i in range(n)]
    out = {"pass1": int(p1["overlaps"].shape[0])}
    # TrimAndAnnotatePiles (construct.cc:123-152)
    begin, end = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    median = np.zeros(n, np.uint16)
    invalid, contained, chimeric = np.zeros(n, bool), np.zeros(n, bool), np.zeros(n, bool)
    regions = [[] for _ in range(n)]
    for i in range(n):
        b, e, m, inv = oracle.pile_trim_and_median(data[i], 4)
        if inv:  # the region (and the median) stay what the constructor set
            begin[i], end[i], invalid[i] = 0, data[i].shape[0], True
            lists[i] = lists[i][:0]
        else:
            begin[i], end[i], median[i] = b, e, m
            regions[i] = [tuple(int(x) for x in r) for r in oracle.find_chimeric_regions(data[i])]
    out["A"] = [(i, int(begin[i]), int(end[i]), int(median[i]), int(invalid[i]), 0, 0, len(regions[i]), data[i].copy())
                for i in range(n)]

    def csr(ls):
        off = np.zeros(n + 1, np.uint32)
        off[1:] = np.cumsum([len(x) for x in ls])
        flat = np.concatenate(ls) if off[-1] else np.zeros(0, oracle.OVERLAP_DTYPE)
        return flat, off

    def bases():
        return begin << 4, end << 4

    # ResolveContainedReads (construct.cc:154-248)
    if identity != 0:
        flat, off = csr(lists)
        bb, ee = bases()
        flat, off = oracle.identity_filter(rs, flat, off, bb, ee, invalid.astype(np.uint8), identity)
        lists = [flat[int(off[i]):int(off[i + 1])].copy() for i in range(n)]
    flat, off = csr(lists)
    bb, ee = bases()
    upd, ok, ty = oracle.overlap_update_and_type(flat, bb, ee, invalid.astype(np.uint8))
    maybe = np.array([len(r) > 0 for r in regions])
    new_lists = []
    for i in range(n):
        keep = []
        for j in range(int(off[i]), int(off[i + 1])):
            if not ok[j]:
                continue
            rhs = int(upd[j]["rhs_id"])
            if ty[j] == 1 and not maybe[rhs]:
                contained[i] = True
            elif ty[j] == 2 and not maybe[i]:
                contained[rhs] = True
            else:
                keep.append(upd[j])
        new_lists.append(np.array(keep, dtype=oracle.OVERLAP_DTYPE) if keep else np.zeros(0, oracle.OVERLAP_DTYPE))
    lists = new_lists
    for i in range(n):
        if contained[i]:
            invalid[i] = True
            lists[i] = lists[i][:0]
    out["resolved"] = int(sum(len(x) for x in lists))
    # ResolveChimericSequences (construct.cc:250-313)
    meds = np.sort(median[median != 0])
    if meds.shape[0]:
        med = int(np.partition(median[median != 0], meds.shape[0] // 2)[meds.shape[0] // 2])
        for i in range(n):
            if invalid[i]:
                continue
            b, e, inv, chim, unres = _clear_chimeric_regions(data[i], int(begin[i]), int(end[i]), regions[i], med)
            begin[i], end[i], regions[i] = b, e, unres
            chimeric[i] = chim
            if inv:
                invalid[i] = True
                lists[i] = lists[i][:0]
        flat, off = csr(lists)
        bb, ee = bases()
        upd, ok, ty = oracle.overlap_update_and_type(flat, bb, ee, invalid.astype(np.uint8))
        for j in range(flat.shape[0]):
            if not ok[j]:
                continue
            if ty[j] == 1:
                contained[int(upd[j]["lhs_id"])] = invalid[int(upd[j]["lhs_id"])] = True
            elif ty[j] == 2:
                contained[int(upd[j]["rhs_id"])] = invalid[int(upd[j]["rhs_id"])] = True
    out["B"] = [(i, int(begin[i]), int(end[i]), int(median[i]), int(invalid[i]), int(contained[i]), int(chimeric[i]),
                 len(regions[i]), data[i].copy()) for i in range(n)]
    # stage -4: FindOverlapsAndRepetetiveRegions (construct.cc:316-491)
    bb, ee = bases()
    out["pass2"] = oracle.second_pass(15, 5, rs, bb, ee, invalid.astype(np.uint8), freq=0.001, kmer_len=15, identity=identity)
    out["contained_before"] = contained.copy()
    out["invalid_before"] = invalid.copy()
    return out


def _hash(values):
    h = 0
    for v in values.tolist():
        h = (h * 1000003 + v) & 0xFFFFFFFFFFFFFFFF
    return h


def _chimeric_reads(seed, read_len):
    g = synth.make_genome(12 * read_len, seed=seed)
    rs, _ = synth.make_reads(g, 16, read_len, seed=seed + 1)  # ~190 reads
    codes = [rs.codes(i) for i in range(rs.n)]
    for i in range(0, rs.n - 1, 9):  # every 9th read chimeric: two distant segments joined
        codes[i] = np.concatenate([codes[i][:len(codes[i]) // 2], codes[(i + rs.n // 2) % rs.n][:read_len // 2]])
    for i in range(4, rs.n, 11):  # and some reads cut short: contained in their neighbours
        codes[i] = codes[i][read_len // 5:read_len // 5 + read_len // 2]
    return seqio.pack_reads(codes)


@pytest.mark.parametrize("identity", [0.0, 0.78])
def test_construct_stages_match_the_oracles_statement_of_the_same_sequence(tmp_path, identity):
    exe = _build(tmp_path, "construct_stage_test")
    # with the identity filter the oracle's checker aligns every overlap with a quadratic DP: shorter reads there
    rs = _chimeric_reads(301, 5000 if identity == 0 else 3000)
    path = _write_reads(tmp_path, rs)
    r = subprocess.run([exe, "stages", path, str(identity)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().split("\n")
    want = _oracle_stages(rs, identity)
    assert lines[0] == "pass1 overlaps %d" % want["pass1"]
    for tag in ("A", "B"):
        got = [ln for ln in lines if ln.startswith(tag + " ")]
        exp = ["%s %d %d %d %d %d %d %d %d %d" % ((tag,) + t[:8] + (_hash(t[8]),)) for t in want[tag]]
        assert got == exp, tag
    assert "resolved overlaps %d" % want["resolved"] in lines
    # both outcomes of every decision occur on this read set
    b = want["B"]
    assert sum(t[4] for t in b) > 5 and sum(1 - t[4] for t in b) > 10          # invalid / valid piles
    assert sum(t[5] for t in b) > 3                                            # contained
    # (chimeric junctions of this read set are cut by FindValidRegion(4) already; ClearChimericRegions' own branches
    # are exercised by the equality of the "B" dump whenever regions survive the trim)
    p2 = want["pass2"]
    assert "lists %d" % (rs.n + 1) in lines and lines[-1] == "stage -3"
    got_o = [ln for ln in lines if ln.startswith("O ")]
    exp_o = ["O %d %d %d %d %d %d %d %d" % (o["lhs_id"], o["lhs_begin"], o["lhs_end"], o["rhs_id"], o["rhs_begin"],
                                             o["rhs_end"], o["score"], 1 if o["strand"] else 0) for o in p2["overlaps"]]
    assert len(exp_o) > 20 and got_o == exp_o
    for i in range(rs.n):
        c = bool(want["contained_before"][i]) or bool(p2["contained"][i])
        inv = bool(want["invalid_before"][i]) or bool(p2["contained"][i])
        k = p2["kmers"][i]
        assert "K %d %d %d %d %d" % (i, c, inv, k.shape[0], _hash(k)) in lines, i


def test_unitig_names_through_the_polisher_rounds(tmp_path):
    """GetUnitigs' `Utg<node> LN:i: RC:i: XO:i:` names feed racon::Polisher; raven::Polish reads the node id after "Utg"
    and the polished ratio after the last ':' back, rotates circular unitigs by 0.42 and feeds the result to the next
    round (polish.cc:50-74).  Against the same two rounds through the ctypes path."""
    exe = _build(tmp_path, "construct_stage_test")
    g = synth.make_genome(36_000, seed=77)
    truths = [g[:14_000], g[14_000:26_000], g[26_000:36_000]]
    drafts = [synth.make_draft(t, seed=80 + i) for i, t in enumerate(truths)]
    rs, _ = synth.make_reads(g, 20, 3000, seed=78)
    dpath = str(tmp_path / "drafts.txt")
    with open(dpath, "wb") as f:
        for d in drafts:
            f.write(bytes(np.frombuffer(b"ACGT", np.uint8)[d]) + b"\n")
    rpath = _write_reads(tmp_path, rs)
    r = subprocess.run([exe, "polish", dpath, rpath, "2"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().split("\n")
    eng = hip.Engine(15, 5)
    reads = eng.upload(rs)
    cur = [d.copy() for d in drafts]
    circular = [False, True, False]
    for rnd in range(2):
        cons, ratio, _ = eng.polish_round(eng.upload(seqio.pack_reads(cur)), reads)
        names = [ln for ln in lines if ln.startswith("R %d " % rnd)]
        assert len(names) == 3
        for i in range(3):
            node = 100 + 2 * i
            assert names[i].startswith("R %d Utg%d LN:i:%d RC:i:" % (rnd, node, len(cons[i])))
            assert abs(float(names[i].rsplit(":", 1)[1]) - ratio[i]) < 1e-6
            c = cons[i]
            if circular[i] and ratio[i] > 0:
                b = int(0.42 * len(c))
                c = np.concatenate([c[b:], c[:b]])
            cur[i] = c if ratio[i] > 0 else cur[i]
    for i in range(3):
        want = bytes(np.frombuffer(b"ACGT", np.uint8)[cur[i]]).decode()
        assert "N %d 1 %s" % (100 + 2 * i, want) in lines


def test_salvage_plasmids_call_sequence(tmp_path):
    """assemble.cc:732-795: Minimize(plasmids) / Filter / Map(it, true, true) among the circular non-unitig sequences,
    then Minimize(unitigs, minhash) / Filter / Map(it, false, false) of sequences that are NOT in the index."""
    exe = _build(tmp_path, "construct_stage_test")
    rng = np.random.default_rng(9)
    g = synth.make_genome(80_000, seed=5)
    uni = [g[:50_000], g[50_000:]]
    p_new = rng.integers(0, 4, 6000, dtype=np.uint8)          # a plasmid found nowhere else: kept
    p_dup = synth.mutate(rng, p_new, 0.01, 0.005, 0.005)       # near copy of it: duplicate within the plasmids
    p_in_unitig = synth.mutate(rng, g[10_000:17_000], 0.01, 0.005, 0.005)  # part of a unitig: duplicate of a unitig
    p_other = rng.integers(0, 4, 3000, dtype=np.uint8)         # another novel one: kept
    plasmids = [p_new, p_in_unitig, p_dup, p_other]

    def dump(path, seqs):
        with open(path, "wb") as f:
            for s in seqs:
                f.write(bytes(np.frombuffer(b"ACGT", np.uint8)[s]) + b"\n")
    ppath, upath = str(tmp_path / "pl.txt"), str(tmp_path / "un.txt")
    dump(ppath, plasmids)
    dump(upath, uni)
    r = subprocess.run([exe, "plasmids", ppath, upath], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    # the same decisions with the oracle's engine: sorted by length, ids = positions
    order = sorted(range(4), key=lambda i: len(plasmids[i]))
    prs = seqio.pack_reads([plasmids[i] for i in order])
    oe = oracle.Engine(15, 5)
    oe.minimize(prs, minhash=False)
    oe.filter(0.001)
    dup_within = [len(oe.map(prs, i, avoid_equal=True, avoid_symmetric=True)["overlaps"]) > 0 for i in range(4)]
    left = [i for i in range(4) if not dup_within[i]]
    urs = seqio.pack_reads(uni)
    oe2 = oracle.Engine(15, 5)
    oe2.minimize(urs, minhash=True)
    oe2.filter(0.001)
    both = seqio.pack_reads(uni + [plasmids[order[i]] for i in left])  # queries that are not in the index
    dup_unitig = [len(oe2.map(both, 2 + x, avoid_equal=False, avoid_symmetric=False)["overlaps"]) > 0 for x in range(len(left))]
    kept = [order[i] for x, i in enumerate(left) if not dup_unitig[x]]
    assert sum(dup_within) == 1 and sum(dup_unitig) == 1 and len(kept) == 2  # every branch taken
    assert lines[-1] == "salvaged %d" % len(kept)
    assert sorted(ln.split()[1] for ln in lines if ln.startswith("kept ")) == sorted("Ctg%d" % i for i in kept)
    assert [ln.split()[1] for ln in lines if ln.startswith("dup_within ")] == ["Ctg%d" % order[i] for i in range(4) if dup_within[i]]
    assert [ln.split()[1] for ln in lines if ln.startswith("dup_unitig ")] == ["Ctg%d" % order[i] for x, i in enumerate(left) if dup_unitig[x]]
