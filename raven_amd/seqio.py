"""Host-side sequence plumbing of the ctypes binding: biosoup-style 2-bit packing of in-memory sequences.

Files are read by the library itself (rvn_reads_load, raven_amd/csrc/io.hip); the independent Python FASTA / FASTQ
parser the tests check it against lives in oracle/seqio_oracle.py (test infrastructure).
Mirrors biosoup::NucleicAcid: 32 bases per uint64, base i at bits (2i mod 64) LSB-first,
A=0 C=1 G=2 T=3, IUPAC folded to 0..3, anything else -> ValueError
(biosoup throws std::invalid_argument).  Every read starts on a word boundary.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# biosoup coder table
_CODER = np.full(256, 255, dtype=np.uint8)
for _chars, _code in (("AaDdNnRrWw-", 0), ("CcBbMmSs", 1), ("GgKkVv", 2), ("TtUuHhYy", 3)):
    for _c in _chars:
        _CODER[ord(_c)] = _code

_DECODER = np.frombuffer(b"ACGT", dtype=np.uint8)


@dataclass
class ReadSet:
    """Concatenated 2-bit packed reads (host memory)."""

    packed: np.ndarray  # uint64 words, all reads concatenated
    word_offsets: np.ndarray  # uint64[n+1], word index where read i starts
    lengths: np.ndarray  # uint32[n]
    ids: np.ndarray  # uint32[n]
    names: list | None = None
    qualities: list | None = None  # optional list of uint8 arrays (Phred+33 removed)

    @property
    def n(self) -> int:
        return int(self.lengths.shape[0])

    @property
    def total_bases(self) -> int:
        return int(self.lengths.astype(np.uint64).sum())

    def codes(self, i: int) -> np.ndarray:
        """2-bit codes of read i as uint8 array."""
        w = self.packed[int(self.word_offsets[i]): int(self.word_offsets[i + 1])]
        n = int(self.lengths[i])
        shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))
        c = ((w[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)
        return c[:n]

    def inflate(self, i: int) -> bytes:
        return _DECODER[self.codes(i)].tobytes()


def encode(seq: bytes | str) -> np.ndarray:
    if isinstance(seq, str):
        seq = seq.encode()
    codes = _CODER[np.frombuffer(seq, dtype=np.uint8)]
    if codes.size and codes.max() == 255:
        raise ValueError("[raven_amd::seqio] error: invalid character in sequence")
    return codes


def pack_codes(codes: np.ndarray) -> np.ndarray:
    """uint8 codes (0..3) -> uint64 words, LSB-first."""
    n = codes.shape[0]
    nw = (n + 31) // 32
    buf = np.zeros(nw * 32, dtype=np.uint64)
    buf[:n] = codes
    buf = buf.reshape(nw, 32)
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))
    return np.bitwise_or.reduce(buf << shifts[None, :], axis=1)


def pack_reads(code_arrays, ids=None, names=None, qualities=None) -> ReadSet:
    lengths = np.array([c.shape[0] for c in code_arrays], dtype=np.uint32)
    nwords = (lengths.astype(np.uint64) + np.uint64(31)) // np.uint64(32)
    word_offsets = np.zeros(len(code_arrays) + 1, dtype=np.uint64)
    np.cumsum(nwords, out=word_offsets[1:])
    packed = np.zeros(int(word_offsets[-1]) + 1, dtype=np.uint64)  # +1 pad word for device over-read
    for i, c in enumerate(code_arrays):
        packed[int(word_offsets[i]): int(word_offsets[i + 1])] = pack_codes(c)
    if ids is None:
        ids = np.arange(len(code_arrays), dtype=np.uint32)
    return ReadSet(packed, word_offsets, lengths, np.asarray(ids, dtype=np.uint32), names, qualities)


def bgzf_compress(data: bytes, level: int = 1, block: int = 65280) -> bytes:
    """Blocked gzip as bgzip writes it (SAM specification 4.1: every member carries its own size in a 'BC' extra
    subfield, an empty member ends the file): the multi-member form rvn_reads_load inflates with a pool of threads."""
    import struct
    import zlib
    out = []
    for a in list(range(0, len(data), block)) + [None]:
        chunk = b"" if a is None else data[a:a + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(body) + 25)
                   + body + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out)
