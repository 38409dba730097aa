"""TEST INFRASTRUCTURE — independent Python FASTA / FASTQ(.gz) parser.

The product reads files with rvn_reads_load (raven_amd/csrc/io.hip: bioparser's role, RavenLib/src/io.cc:7-41); this
is the checker the tests and the golden-fixture generators compare it with, restated from the file formats alone (no
code shared with the library).  Only tests/, tools/ and the fixture generators import it.
"""
from __future__ import annotations

import gzip

import numpy as np

from raven_amd.seqio import ReadSet, encode, pack_reads


def _open(path):
    return gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")


def parse_fastq(path, limit: int | None = None):
    names, seqs, quals = [], [], []
    with _open(path) as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip(b"\r\n")
            f.readline()
            q = f.readline().rstrip(b"\r\n")
            names.append(h[1:].split()[0].decode())
            seqs.append(s)
            quals.append(np.frombuffer(q, dtype=np.uint8) - 33)
            if limit is not None and len(seqs) >= limit:
                break
    return names, seqs, quals


def parse_fasta(path, limit: int | None = None):
    names, seqs = [], []
    cur = []
    with _open(path) as f:
        for line in f:
            if line.startswith(b">"):
                if cur or names:
                    seqs.append(b"".join(cur))
                    cur = []
                    if limit is not None and len(seqs) >= limit:
                        names = names[:limit]
                        return names, seqs
                names.append(line[1:].split()[0].decode())
            else:
                cur.append(line.strip())
    if names:
        seqs.append(b"".join(cur))
    return names, seqs


def load_reads(path, limit: int | None = None) -> ReadSet:
    """Extension sniffing as in io.cc:7-41."""
    p = str(path)
    base = p[:-3] if p.endswith(".gz") else p
    if base.endswith((".fastq", ".fq")):
        names, seqs, quals = parse_fastq(p, limit)
    elif base.endswith((".fasta", ".fa")):
        names, seqs = parse_fasta(p, limit)
        quals = None
    else:
        raise ValueError(
            "[raven_amd::seqio] error: file %s has unsupported format extension "
            "(valid extensions: .fasta, .fasta.gz, .fa, .fa.gz, .fastq, .fastq.gz, .fq, .fq.gz)" % p)
    return pack_reads([encode(s) for s in seqs], names=names, qualities=quals)
