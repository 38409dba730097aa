"""The input path (rvn_reads_load: gz FASTA / FASTQ -> inflate pool -> page-locked text slabs -> record scanner -> text
in HBM -> 2-bit packing and block qualities on the device; RavenLib/src/io.cc:7-41 + biosoup::NucleicAcid) against the Python restatement of the same
rules (raven_amd/seqio.py) on the reference's own data files and on synthetic multi-chunk files."""
import gzip
import os

import numpy as np
import pytest

from raven_amd import hip, seqio, synth
from oracle import seqio_oracle

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same_reads(rd, rs):
    packed, woff, lens, q, qoff, shift = rd.fetch()
    assert np.array_equal(lens, rs.lengths) and np.array_equal(woff, rs.word_offsets)
    assert np.array_equal(packed, rs.packed[: int(rs.word_offsets[-1])])
    return q, qoff, shift


def test_lambda_fastq_gz_matches_python_loader():
    path = os.path.join(GOLDEN, "ERA476754.fastq.gz")
    rs = seqio_oracle.load_reads(path)
    eng = hip.Engine(15, 5)
    rd = eng.load(path)
    assert rd.n == rs.n == 236 and rd.rs.names == rs.names and rd.load_stats["has_quality"] == 1
    assert rd.load_stats["n_bases"] == rs.total_bases
    q, qoff, shift = _same_reads(rd, rs)
    assert shift == 6
    for i in range(rs.n):  # biosoup block_quality: integer mean of every 64-base block (+33 as stored)
        ph = np.asarray(rs.qualities[i], dtype=np.int64)
        nb = (ph.shape[0] + 63) // 64
        want = np.array([int(ph[b * 64:(b + 1) * 64].sum()) // len(ph[b * 64:(b + 1) * 64]) for b in range(nb)], np.int64) + 33
        assert np.array_equal(q[int(qoff[i]):int(qoff[i + 1])].astype(np.int64), want), i
    # the loaded set is a working read set: the first pass gives what the uploaded one gives
    a = eng.find_overlaps_and_create_piles(rd)
    b = eng.find_overlaps_and_create_piles(eng.upload(rs))
    assert np.array_equal(a.overlaps()[0], b.overlaps()[0]) and np.array_equal(a.piles()[0], b.piles()[0])


def test_lambda_fasta_gz_multi_line_records():
    path = os.path.join(GOLDEN, "NC_001416.fasta.gz")
    rs = seqio_oracle.load_reads(path)
    eng = hip.Engine(15, 5)
    rd = eng.load(path)
    assert rd.n == 1 and rd.rs.names == rs.names and rd.load_stats["has_quality"] == 0
    _, _, shift = _same_reads(rd, rs)
    assert shift == -1


def test_multi_chunk_plain_fasta_and_iupac(tmp_path):
    g = synth.make_genome(400_000, seed=3)
    rs, _ = synth.make_reads(g, 190, 10000, seed=4)  # ~76 MB of bases: more than one 64 MB staging chunk
    path = str(tmp_path / "reads.fa")
    with open(path, "wb") as f:
        for i in range(rs.n):
            s = rs.inflate(i)
            f.write(b">r%d some description\n" % i)
            for x in range(0, len(s), 70000):  # wrapped lines
                f.write(s[x:x + 70000] + b"\r\n")
    eng = hip.Engine(15, 5)
    rd = eng.load(path)
    assert rd.n == rs.n and rd.rs.names == ["r%d" % i for i in range(rs.n)]
    _same_reads(rd, rs)
    # IUPAC codes fold to ACGT exactly as biosoup's coder table does
    iupac = b"ACGTUacgtuNnRrYyKkMmSsWwBbDdHhVv-"
    p2 = str(tmp_path / "iupac.fasta.gz")
    with gzip.open(p2, "wb") as f:
        f.write(b">x\n" + iupac + b"\n>empty\n\n>y\nAC\n")
    rd2 = eng.load(p2)
    want = seqio.pack_reads([seqio.encode(iupac), seqio.encode(b""), seqio.encode(b"AC")])
    assert rd2.n == 3 and rd2.rs.names == ["x", "empty", "y"]
    _same_reads(rd2, want)


def test_errors_are_the_references(tmp_path):
    eng = hip.Engine(15, 5)
    with pytest.raises(ValueError, match="unsupported format extension"):
        eng.load(str(tmp_path / "reads.txt"))
    with pytest.raises(ValueError, match="unable to open"):
        eng.load(str(tmp_path / "missing.fasta"))
    bad = tmp_path / "bad.fa"
    bad.write_bytes(b">a\nACGTXACGT\n")
    with pytest.raises(ValueError, match="not a nucleotide"):
        eng.load(str(bad))
    trunc = tmp_path / "trunc.fastq"
    trunc.write_bytes(b"@a\nACGTACGT\n+\nIIII\n")
    with pytest.raises(ValueError, match="invalid file format"):
        eng.load(str(trunc))
    nofa = tmp_path / "nofa.fasta"
    nofa.write_bytes(b"ACGT\n")
    with pytest.raises(ValueError, match="invalid file format"):
        eng.load(str(nofa))


def test_truncated_or_corrupt_gz_is_an_error_not_a_shorter_read_set(tmp_path):
    """zlib reports a cut or damaged archive through gzread's return value / gzerror: that must surface as the parser's
    error, never as a clean end of file (a truncated .gz used to load as a shorter read set)."""
    import gzip
    eng = hip.Engine(15, 5)
    rng = np.random.default_rng(5)
    recs = []
    for i in range(400):
        seq = "".join("ACGT"[x] for x in rng.integers(0, 4, 700))
        recs.append("@r%d\n%s\n+\n%s\n" % (i, seq, "I" * 700))
    blob = gzip.compress("".join(recs).encode())
    whole = tmp_path / "whole.fastq.gz"
    whole.write_bytes(blob)
    assert eng.load(str(whole)).n == 400
    cut = tmp_path / "cut.fastq.gz"
    cut.write_bytes(blob[:len(blob) // 2])
    with pytest.raises(ValueError, match="corrupt or truncated"):
        eng.load(str(cut))
    bad = bytearray(blob)
    for k in range(len(bad) // 2, len(bad) // 2 + 64):
        bad[k] ^= 0x5A
    dmg = tmp_path / "damaged.fasta.gz"
    dmg.write_bytes(bytes(bad))
    with pytest.raises(ValueError, match="corrupt or truncated|invalid file format|not a nucleotide"):
        eng.load(str(dmg))


def _fastq_bytes(rs, qual_of):
    out = []
    for i in range(rs.n):
        sq = rs.inflate(i)
        out.append(b"@r%d\n" % i + sq + b"\n+\n" + qual_of(i, len(sq)) + b"\n")
    return b"".join(out)


def test_bgzf_fastq_many_slabs_two_device_batches(tmp_path):
    """~90 Mbase of FASTQ (180 MB of text: more than one 128 MB device batch, a few dozen slabs) as BGZF: the pool inflates
    the blocks in parallel, reads and block qualities equal the Python restatement; the same text as ONE gzip member goes
    front to back on one thread and gives the same read set."""
    from tests.test_io_text import _bgzf
    g = synth.make_genome(600_000, seed=13)
    rs, _ = synth.make_reads(g, 150, 15000, seed=14)
    rng = np.random.default_rng(15)
    quals = [bytes(rng.integers(33, 74, int(n)).astype(np.uint8)) for n in rs.lengths]
    text = _fastq_bytes(rs, lambda i, n: quals[i])
    assert len(text) > 150 << 20
    eng = hip.Engine(15, 5)
    p = str(tmp_path / "reads.fastq.gz")
    open(p, "wb").write(_bgzf(text))
    rd = eng.load(p)
    st = rd.load_stats
    assert st["streaming"] == 0 and st["restarted"] == 0 and st["members"] > 2000 and st["inflate_threads"] >= 1
    assert rd.n == rs.n and st["n_bases"] == rs.total_bases
    q, qoff, shift = _same_reads(rd, rs)
    assert shift == 6
    for i in (0, 1, rs.n // 2, rs.n - 1):
        a = np.frombuffer(quals[i], dtype=np.uint8).astype(np.int64) - 33
        want = np.array([int(a[x:x + 64].sum()) // len(a[x:x + 64]) + 33 for x in range(0, len(a), 64)], dtype=np.uint8)
        assert np.array_equal(q[int(qoff[i]):int(qoff[i + 1])], want)
    rd.close()
    p1 = str(tmp_path / "one.fastq.gz")
    with gzip.open(p1, "wb", compresslevel=1) as f:
        f.write(text)
    rd1 = eng.load(p1)
    assert rd1.load_stats["streaming"] == 1 and rd1.load_stats["inflate_threads"] == 1
    q1, qoff1, _ = _same_reads(rd1, rs)
    assert np.array_equal(q1, q) and np.array_equal(qoff1, qoff)


def test_chromosome_sized_records_outlive_slabs_and_batches(tmp_path):
    """Records far longer than a slab and than a device batch (a 150 Mb sequence wrapped at 80 columns with CRLF, then a
    30 Mb one on a single line, then short ones): the text of the record in progress is carried from batch to batch."""
    from tests.test_io_text import _bgzf
    rng = np.random.default_rng(21)
    big = rng.integers(0, 4, 150_000_000, dtype=np.uint8)
    mid = rng.integers(0, 4, 30_000_000, dtype=np.uint8)
    small = [rng.integers(0, 4, int(n), dtype=np.uint8) for n in (1, 31, 32, 33, 5000)]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    b = lut[big]
    n_full = len(b) // 80
    wrapped = np.empty((n_full, 82), dtype=np.uint8)
    wrapped[:, :80] = b[:n_full * 80].reshape(n_full, 80)
    wrapped[:, 80] = 13
    wrapped[:, 81] = 10
    text = (b">chr1 wrapped\r\n" + wrapped.tobytes() + lut[big[n_full * 80:]].tobytes() + b"\r\n>chr2\n" + lut[mid].tobytes() + b"\n"
            + b"".join(b">s%d\n" % i + lut[s].tobytes() + b"\n" for i, s in enumerate(small)))
    want = seqio.pack_reads([big, mid] + small)
    eng = hip.Engine(15, 5)
    p = str(tmp_path / "genome.fa")
    open(p, "wb").write(text)
    rd = eng.load(p)
    assert rd.rs.names == ["chr1", "chr2"] + ["s%d" % i for i in range(5)]
    _same_reads(rd, want)
    rd.close()
    p = str(tmp_path / "genome.fa.gz")
    open(p, "wb").write(_bgzf(text))
    rd = eng.load(p)
    _same_reads(rd, want)
