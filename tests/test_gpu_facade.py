"""The C++ facade (include/ram/minimizer_engine.hpp + include/raven_hip/find_overlaps.hpp) compiled with g++
against tests/cpp test doubles of biosoup, run on the GPU and compared with the ctypes path."""
import os
import subprocess

import numpy as np
import pytest

from raven_amd import hip, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, name="facade_test"):
    exe = str(tmp_path / name)
    lib = os.path.join(ROOT, "raven_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp"),
           "-o", exe, os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-L", lib, "-lraven_hip",
           "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def _write_reads(tmp_path, rs):
    path = str(tmp_path / "reads.txt")
    with open(path, "wb") as f:
        for i in range(rs.n):
            f.write(rs.inflate(i) + b"\n")
    return path


def test_facade_compiles_and_fails_loudly_without_gpu(tmp_path):
    if hip.device_count() > 0:
        pytest.skip("GPU present")
    exe = _build(tmp_path)
    p = tmp_path / "one.txt"
    p.write_text("ACGTACGTACGTACGTACGTACGTACGT\n")
    r = subprocess.run([exe, str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_facade_matches_c_abi(tmp_path):
    exe = _build(tmp_path)
    g = synth.make_genome(80_000, seed=41)
    rs, _ = synth.make_reads(g, 12, 5000, seed=42)
    path = _write_reads(tmp_path, rs)
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines[0] == "filter_throws 1"
    eng = hip.Engine(15, 5)
    rd = eng.upload(rs)
    p = eng.find_overlaps_and_create_piles(rd)
    data, _ = p.piles()
    ovl, off = p.overlaps()
    cov = 0
    for v in data.tolist():
        cov = (cov * 1000003 + v) & 0xFFFFFFFFFFFFFFFF
    assert lines[1] == "piles %d cov_hash %d" % (rs.n, cov)
    got = [tuple(int(x) for x in ln.split()[1:]) for ln in lines if ln.startswith("O ")]
    pile_of = np.repeat(np.arange(rs.n), np.diff(off.astype(np.int64)))
    want = [(int(pi), int(o["lhs_id"]), int(o["lhs_begin"]), int(o["lhs_end"]), int(o["rhs_id"]), int(o["rhs_begin"]),
             int(o["rhs_end"]), int(o["score"]), int(o["strand"])) for pi, o in zip(pile_of, ovl)]
    assert got == want and len(got) > 100
    assert lines[-2].startswith("map_single_vs_batch mismatches 0 total ")
    assert int(lines[-2].split()[-1]) > 0
    # Map() called from 8 threads at once (indexed sequences and outsiders) == MapBatch, incl. `filtered`
    assert lines[-1].startswith("map_concurrent mismatches 0 total ") and int(lines[-1].split()[-1]) > 0


def test_polisher_facade_compiles_and_fails_loudly_without_gpu(tmp_path):
    if hip.device_count() > 0:
        pytest.skip("GPU present")
    exe = _build(tmp_path, "polisher_test")
    p = tmp_path / "one.txt"
    p.write_text("ACGTACGTACGTACGTACGTACGTACGT\n")
    r = subprocess.run([exe, str(p), str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("q", [None, 12])
def test_polisher_facade_matches_c_abi(tmp_path, q):
    """racon::Polisher facade == rvn_polish_round through ctypes, incl. the name tags Raven parses."""
    from raven_amd import seqio
    from tests import polish_util
    truths, drafts, targets, reads, _ = polish_util.make_case(genome_len=30_000, coverage=20, read_len=2500, seed=11,
                                                              n_targets=2)
    tpath, rpath = _write_reads(tmp_path, targets), str(tmp_path / "r.txt")
    os.rename(tpath, str(tmp_path / "t.txt"))
    os.rename(_write_reads(tmp_path, reads), rpath)
    exe = _build(tmp_path, "polisher_test")
    r = subprocess.run([exe, str(tmp_path / "t.txt"), rpath] + ([str(q)] if q is not None else []),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines[0] == "zero_window_throws 1" and lines[1] == "polished 2" and lines[-1] == "dropped_without_reads 2"
    eng = hip.Engine(15, 5)
    quals = [np.full(int(n), 33 + q, dtype=np.uint8) for n in reads.lengths] if q is not None else None
    cons, ratio, st = eng.polish_round(eng.upload(targets), eng.upload(reads), quals=quals, q=10.0 if q else 0.0)
    P = [ln.split(" ", 3) for ln in lines if ln.startswith("P ")]
    S = [ln[2:] for ln in lines if ln.startswith("S ")]
    for t in range(2):
        assert int(P[t][1]) == t and abs(float(P[t][2]) - ratio[t]) < 1e-6 and ratio[t] > 0.9
        name = P[t][3]
        assert name.startswith("Utg%d LN:i:%d RC:i:" % (t, len(cons[t]))) and " XC:f:" in name
        assert S[t].encode() == bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[cons[t]])
    used = [int(p[3].split("RC:i:")[1].split()[0]) for p in P]
    assert sum(used) == st["n_reads_used"]
    # second round = polish.cc:50-52 (the result of the first as targets: the facade takes the engine's resident consensus),
    # third round with the first target rotated in place like a circular unitig (polish.cc:60-65: the facade uploads)
    rd = eng.upload(reads)
    c2, _, _ = eng.polish_round(eng.upload_codes(cons), rd, quals=quals, q=10.0 if q else 0.0)
    S2 = [ln[3:] for ln in lines if ln.startswith("S2 ")]
    assert len(S2) == 2 and all(S2[t].encode() == bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[c2[t]]) for t in range(2))
    b = int(0.42 * len(c2[0]))
    c3, _, _ = eng.polish_round(eng.upload_codes([np.concatenate([c2[0][b:], c2[0][:b]]), c2[1]]), rd, quals=quals, q=10.0 if q else 0.0)
    assert "resident_rounds 1" in lines  # (the second round took the engine's copy, the third — rotated — did not)
    S3 = [ln[3:] for ln in lines if ln.startswith("S3 ")]
    assert len(S3) == 2 and all(S3[t].encode() == bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[c3[t]]) for t in range(2))


def test_edlib_dropin_compiles_and_fails_loudly_without_gpu(tmp_path):
    """include/edlib.h: construct.cc:190-199's call pattern compiles; without a GPU edlibAlign reports
    EDLIB_STATUS_ERROR (the reference then scores the overlap 0) instead of computing anything on the CPU."""
    if hip.device_count() > 0:
        pytest.skip("GPU present")
    exe = _build(tmp_path, "edlib_test")
    p = tmp_path / "two.txt"
    p.write_text("ACGTACGTACGTACGTACGTACGTACGT\nACGTACGAACGTACGTACGTACGTACGT\n")
    r = subprocess.run([exe, str(p)], capture_output=True, text=True)
    lines = r.stdout.strip().split("\n")
    assert lines[0] == "default_config -1 0 0" and lines[1] == "hw_mode_status 1" and lines[2] == "five_symbols_status 1"
    assert r.returncode == 1 and lines[3] == "NO_DEVICE status 1"


@pytest.mark.gpu
def test_edlib_dropin_matches_dp(tmp_path):
    """edlibAlign drop-in called from 8 threads (combined into device batches) == textbook DP, both strands."""
    exe = _build(tmp_path, "edlib_test")
    g = synth.make_genome(3000, seed=5)
    rng = np.random.default_rng(6)
    seqs = []
    for i in range(41):
        a = int(rng.integers(0, 800))
        piece = synth.mutate(rng, g[a:a + int(rng.integers(300, 2200))], 0.04, 0.03, 0.03)
        if i % 7 == 3:
            piece = piece[:int(rng.integers(0, 5))]  # empty / tiny sequences
        seqs.append(bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[piece]))
    path = tmp_path / "seqs.txt"
    path.write_bytes(b"\n".join(seqs) + b"\n")
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines[3].startswith("self 0 locations 1 end %d alphabet " % (len(seqs[0]) - 1))
    assert lines[4] == "pairs 40 mismatches 0"
    assert lines[5] == "k_below -1" and lines[6] == "k_equal 1"


@pytest.mark.gpu
def test_second_pass_facade_matches_c_abi(tmp_path):
    """raven::FindOverlapsAndRepetetiveRegions<Pile> (include/raven_hip/find_overlaps.hpp) with the reference's
    signature, after the first-pass template, against the ctypes path of the same two C-ABI calls."""
    exe = _build(tmp_path, "pass2_facade_test")
    g = synth.make_genome(60_000, seed=51)
    rs, _ = synth.make_reads(g, 14, 4000, seed=52)
    path = _write_reads(tmp_path, rs)
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    eng = hip.Engine(15, 5)
    rd = eng.upload(rs)
    eng.find_overlaps_and_create_piles(rd).close()  # the facade ran the first pass before (same engine state order)
    begin = np.zeros(rs.n, dtype=np.uint32)
    end = (rs.lengths.astype(np.uint32) >> 4) << 4
    invalid = np.zeros(rs.n, np.uint8)
    invalid[-1] = 1  # as the program does: without any invalid pile the reference's loop maps nothing
    res = eng.find_overlaps_and_repetitive_regions(rd, begin, end, invalid, kmer_len=28)
    assert lines[0] == "lists %d" % (rs.n + 1)
    want = ["O %d %d %d %d %d %d %d %d" % (o["lhs_id"], o["lhs_begin"], o["lhs_end"], o["rhs_id"], o["rhs_begin"],
                                            o["rhs_end"], o["score"], 1 if o["strand"] else 0) for o in res["overlaps"]]
    got = [ln for ln in lines if ln.startswith("O ")]
    assert len(want) > 50 and got == want
    for i in range(rs.n):
        h = 0
        for v in res["kmers"][i].tolist():
            h = (h * 1000003 + v) & 0xFFFFFFFFFFFFFFFF
        c = int(res["contained"][i])
        assert "P %d %d %d %d %d" % (i, c, 1 if (c or invalid[i]) else 0, res["kmers"][i].shape[0], h) in lines
