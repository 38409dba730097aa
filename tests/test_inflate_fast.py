"""The single-stream deflate decoder of the input path (raven_amd/csrc/inflate_fast.h through rvn_test_inflate_fast of
libraven_hip_test.so; no GPU) against zlib: every block type (stored, fixed, dynamic), every compression level and
strategy, data from all-equal bytes to noise, the reference's own lambda files, one-shot and through a small drained
buffer (matches reaching back across the drain), and damaged / truncated members, which must be refused (or end with a
CRC / length that does not match: the caller's check)."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

from raven_amd import hip

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _inflate(member: bytes, cap: int, chunk: int = 0):
    T = hip.test_lib()
    src = np.frombuffer(member, dtype=np.uint8)
    dst = np.zeros(cap + 512, dtype=np.uint8)
    out = np.zeros(4, dtype=np.uint64)
    rc = T.rvn_test_inflate_fast(src.ctypes.data_as(C.c_void_p), len(member), dst.ctypes.data_as(C.c_void_p), cap + 512, chunk,
                                 out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(T.rvn_last_error().decode(errors="replace"))
    return dst[:int(out[0])].tobytes(), int(out[1]), int(out[2]), int(out[3])


def _member(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
    return co.compress(data) + co.flush()


def _check(data: bytes, **kw):
    m = _member(data, **kw)
    for chunk in (0, 70_000, 1_000):
        got, used, crc, isize = _inflate(m, len(data), chunk)
        assert got == data, (len(data), kw, chunk)
        assert used == len(m) and crc == zlib.crc32(data) and isize == len(data) % (1 << 32)


def _datasets():
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    fastq = b"".join(b"@read%d\n" % i + acgt[rng.integers(0, 4, n)].tobytes() + b"\n+\n" + bytes(rng.integers(33, 74, n).astype(np.uint8)) + b"\n"
                     for i, n in enumerate(rng.integers(50, 30000, 60)))
    return {
        "empty": b"",
        "one": b"A",
        "zeros": bytes(300_000),
        "run_pairs": b"AB" * 150_000,
        "period7": b"ACGTACG" * 50_000,
        "noise": bytes(rng.integers(0, 256, 400_000).astype(np.uint8)),
        "two_symbols": bytes(rng.integers(0, 2, 200_000).astype(np.uint8)),
        "skewed": bytes(np.minimum(255, rng.geometric(0.02, 300_000)).astype(np.uint8)),
        "fastq": fastq,
        "text": (b"the quick brown fox jumps over the lazy dog; " * 3000) + bytes(rng.integers(32, 127, 50_000).astype(np.uint8)),
    }


@pytest.mark.parametrize("name", list(_datasets().keys()))
def test_every_level_and_strategy_round_trips(name):
    data = _datasets()[name]
    for level in (0, 1, 4, 6, 9):
        _check(data, level=level)
    _check(data, level=6, strategy=zlib.Z_FIXED)         # fixed Huffman blocks
    _check(data, level=6, strategy=zlib.Z_HUFFMAN_ONLY)  # literals only: an empty distance alphabet
    _check(data, level=6, strategy=zlib.Z_RLE)           # distance 1 only: a one-code distance alphabet
    _check(data, level=9, mem=1)                         # many small dynamic blocks


def test_long_codes_need_subtables():
    """A literal alphabet with frequencies falling off geometrically gets codes of up to 15 bits: beyond the 11-bit table."""
    rng = np.random.default_rng(5)
    sym = np.minimum(255, rng.geometric(0.35, 2_000_000) - 1).astype(np.uint8)
    rare = rng.integers(0, 256, 3000).astype(np.uint8)
    data = np.concatenate([sym, rare, sym[::-1]]).tobytes()
    _check(data, level=6, strategy=zlib.Z_HUFFMAN_ONLY)
    _check(data, level=9)


def test_golden_lambda_files():
    for name in ("ERA476754.fastq.gz", "NC_001416.fasta.gz"):
        blob = open(os.path.join(GOLDEN, name), "rb").read()
        text = gzip.decompress(blob)
        got, used, crc, isize = _inflate(blob, len(text))
        assert got == text and crc == zlib.crc32(text) and isize == len(text)
        got, _, _, _ = _inflate(blob, len(text), 65_536)
        assert got == text


def test_damaged_and_truncated_members_are_refused_or_fail_the_checksum():
    rng = np.random.default_rng(23)
    data = _datasets()["fastq"]
    m = _member(data, level=6)
    refused = wrong = 0
    for trial in range(200):
        bad = bytearray(m)
        k = int(rng.integers(12, len(m) - 8))
        bad[k] ^= 1 << int(rng.integers(0, 8))
        try:
            got, used, crc, isize = _inflate(bytes(bad), len(data) * 2 + 100_000)
        except ValueError:
            refused += 1
            continue
        if got == data:
            continue  # (a flipped bit in a stored block's padding or the like)
        assert crc != zlib.crc32(got) or isize != len(got) % (1 << 32) or used != len(m)
        wrong += 1
    assert refused + wrong >= 190
    for cut in (len(m) // 3, len(m) // 2, len(m) - 9, len(m) - 5):
        with pytest.raises(ValueError):
            _inflate(m[:cut], len(data) + 100_000)
    with pytest.raises(ValueError):  # a distance that reaches in front of the output
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, b"ACGT" * 100)  # preset dictionary, then dropped
        body = co.compress(b"ACGT" * 100 + b"TTTT") + co.flush()
        _inflate(b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03" + body + b"\0" * 8, 10_000)
