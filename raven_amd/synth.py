"""Seeded synthetic genomes and long reads (BASELINE.json configs).

Genome: iid uniform ACGT.  ONT-like reads: source segment sampled uniformly,
strand Bernoulli(1/2), iid per-base errors (default 4 % sub, 3 % ins, 3 % del).
Everything is numpy-vectorised in chunks of reads so the 1.5e8-base C2 set is
generated in seconds.  Returns a packed `ReadSet` plus the truth table
(start, source length, strand) used by sanity tests.
"""
from __future__ import annotations

import numpy as np

from .seqio import ReadSet


def make_genome(n_bases: int, seed: int = 0x5EED0001) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 4, size=n_bases, dtype=np.uint8)


def _pack_many(codes: np.ndarray, lengths: np.ndarray):
    """Pack concatenated codes of many reads, each read word-aligned."""
    n = lengths.shape[0]
    nwords = (lengths.astype(np.int64) + 31) // 32
    word_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(nwords, out=word_off[1:])
    base_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lengths.astype(np.int64), out=base_off[1:])
    # destination slot (in bases, word-aligned per read) of every base
    read_of = np.repeat(np.arange(n, dtype=np.int64), lengths.astype(np.int64))
    within = np.arange(codes.shape[0], dtype=np.int64) - base_off[read_of]
    slot = word_off[read_of] * 32 + within
    buf = np.zeros(int(word_off[-1]) * 32, dtype=np.uint64)
    buf[slot] = codes
    buf = buf.reshape(-1, 32)
    shifts = np.arange(32, dtype=np.uint64) * np.uint64(2)
    words = np.bitwise_or.reduce(buf << shifts[None, :], axis=1)
    return words, word_off


def make_reads(genome: np.ndarray, coverage: float, read_len: int = 10000, *, length_model: str = "fixed",
               sub: float = 0.04, ins: float = 0.03, dele: float = 0.03, seed: int = 0x5EED0002,
               chunk_reads: int = 512, min_len: int = 1000, max_len: int = 60000, sigma: float = 0.5):
    """Returns (ReadSet, truth) with truth = dict(start, src_len, strand)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    G = genome.shape[0]
    target = int(coverage * G)
    if length_model == "fixed":
        n_reads = max(1, target // read_len)
        src_len = np.full(n_reads, min(read_len, G), dtype=np.int64)
    elif length_model == "lognormal":
        lens = []
        tot = 0
        while tot < target:
            L = rng.lognormal(np.log(read_len), sigma, size=4096)
            L = np.clip(L, min_len, min(max_len, G)).astype(np.int64)
            lens.append(L)
            tot += int(L.sum())
        src_len = np.concatenate(lens)
        n_reads = int(np.searchsorted(np.cumsum(src_len), target) + 1)
        src_len = src_len[:n_reads]
    else:
        raise ValueError(length_model)
    start = (rng.random(n_reads) * (G - src_len + 1)).astype(np.int64)
    strand = rng.integers(0, 2, size=n_reads, dtype=np.uint8)

    packed_chunks, lengths_all = [], []
    word_total = 0
    word_offsets = [np.zeros(1, dtype=np.int64)]
    for c0 in range(0, n_reads, chunk_reads):
        c1 = min(n_reads, c0 + chunk_reads)
        sl = src_len[c0:c1]
        m = c1 - c0
        off = np.zeros(m + 1, dtype=np.int64)
        np.cumsum(sl, out=off[1:])
        tot = int(off[-1])
        read_of = np.repeat(np.arange(m, dtype=np.int64), sl)
        within = np.arange(tot, dtype=np.int64) - off[read_of]
        fwd = strand[c0:c1][read_of] == 0
        gpos = np.where(fwd, start[c0:c1][read_of] + within, start[c0:c1][read_of] + sl[read_of] - 1 - within)
        base = genome[gpos]
        base = np.where(fwd, base, 3 - base).astype(np.uint8)
        u = rng.random(tot)
        is_del = u < dele
        is_sub = (~is_del) & (u < dele + sub)
        base = np.where(is_sub, (base + rng.integers(1, 4, size=tot, dtype=np.uint8)) & 3, base).astype(np.uint8)
        n_ins = (rng.random(tot) < ins).astype(np.int64)
        emit = (~is_del).astype(np.int64) + n_ins  # bases emitted per source base
        out_read = np.repeat(read_of, emit)
        out_codes = np.repeat(base, emit)
        # the inserted base is the second copy whenever a kept base also inserts, or the only copy when deleted
        out_off = np.zeros(tot + 1, dtype=np.int64)
        np.cumsum(emit, out=out_off[1:])
        ins_slot = out_off[1:][n_ins > 0] - 1
        out_codes[ins_slot] = rng.integers(0, 4, size=ins_slot.shape[0], dtype=np.uint8)
        lengths = np.bincount(out_read, minlength=m).astype(np.uint32)
        words, woff = _pack_many(out_codes, lengths)
        packed_chunks.append(words)
        lengths_all.append(lengths)
        word_offsets.append(woff[1:] + word_total)
        word_total += int(woff[-1])

    packed = np.concatenate(packed_chunks + [np.zeros(1, dtype=np.uint64)])  # +1 pad word
    lengths = np.concatenate(lengths_all)
    woffs = np.concatenate(word_offsets).astype(np.uint64)
    rs = ReadSet(packed, woffs, lengths, np.arange(n_reads, dtype=np.uint32))
    truth = dict(start=start, src_len=src_len, strand=strand)
    return rs, truth


def mutate(rng, codes, sub, ins, dele):
    """iid substitutions / insertions / deletions on a code array (vectorised)."""
    L = codes.shape[0]
    u = rng.random(L)
    keep = u >= dele
    base = codes.copy()
    s = (u >= dele) & (u < dele + sub)
    base[s] = (base[s] + rng.integers(1, 4, size=int(s.sum()))) & 3
    insm = rng.random(L) < ins
    emit = keep.astype(np.int64) + insm
    seq = np.repeat(base, emit)
    off = np.cumsum(emit)
    slots = off[insm] - 1
    seq[slots] = rng.integers(0, 4, size=slots.shape[0])
    return seq.astype(np.uint8)


def make_draft(genome, seed, sub=0.01, ins=0.008, dele=0.008):
    """An unpolished assembly of `genome` (what raven's layout hands to racon): the truth with iid errors."""
    return mutate(np.random.default_rng(seed), genome, sub, ins, dele)


# ---- the same generator on torch tensors (GPU when available): the 100 Mb / 3 Gbase configs in seconds ----------------
# Plumbing for bench.py and the full-size tests only (torch is used as a random-number / scatter engine; nothing of
# the product depends on it).  Same model as make_reads — uniform start, Bernoulli strand, iid substitutions /
# insertions / deletions — but a different random stream: torch's generator instead of numpy's PCG64.

def make_genome_torch(n_bases: int, seed: int = 0x5EED0001, device="cuda"):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(0, 4, (n_bases,), dtype=torch.uint8, device=device, generator=g)


def mutate_torch(codes, sub, ins, dele, seed):
    """iid errors on a uint8 code tensor (the draft assembly handed to the polishing rounds)."""
    import torch
    dev = codes.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    L = codes.shape[0]
    u = torch.rand(L, device=dev, generator=g)
    keep = u >= dele
    s = keep & (u < dele + sub)
    base = torch.where(s, (codes + torch.randint(1, 4, (L,), dtype=torch.uint8, device=dev, generator=g)) & 3, codes)
    insm = torch.rand(L, device=dev, generator=g) < ins
    emit = keep.to(torch.int64) + insm.to(torch.int64)
    seq = torch.repeat_interleave(base, emit)
    off = torch.cumsum(emit, 0)
    slots = off[insm] - 1
    seq[slots] = torch.randint(0, 4, (int(slots.shape[0]),), dtype=torch.uint8, device=dev, generator=g)
    return seq


def pack_torch(codes, lengths):
    """2-bit pack concatenated codes of many reads, every read word-aligned: (words int64 tensor, word_off int64)."""
    import torch
    dev = codes.device
    lengths = lengths.to(torch.int64)
    n = lengths.shape[0]
    nwords = (lengths + 31) // 32
    word_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(nwords, 0, out=word_off[1:])
    base_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lengths, 0, out=base_off[1:])
    read_of = torch.repeat_interleave(torch.arange(n, device=dev), lengths)
    within = torch.arange(codes.shape[0], device=dev) - base_off[read_of]
    slot = word_off[read_of] * 32 + within
    buf = torch.zeros(int(word_off[-1]) * 32, dtype=torch.int64, device=dev)
    buf[slot] = codes.to(torch.int64)
    shifts = torch.arange(32, device=dev, dtype=torch.int64) * 2
    words = (buf.view(-1, 32) << shifts[None, :]).sum(dim=1)  # disjoint bit fields: the sum is the bitwise or
    return words, word_off


def make_reads_torch(genome, coverage: float, read_len: int = 10000, *, length_model: str = "fixed", sub: float = 0.04,
                     ins: float = 0.03, dele: float = 0.03, seed: int = 0x5EED0002, chunk_bases: int = 1 << 27,
                     min_len: int = 1000, max_len: int = 60000, sigma: float = 0.5):
    """make_reads on the device of `genome` (a uint8 torch tensor).  Returns (ReadSet with numpy host arrays, truth)."""
    import torch
    dev = genome.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    G = int(genome.shape[0])
    target = int(coverage * G)
    if length_model == "fixed":
        n_reads = max(1, target // read_len)
        src_len = torch.full((n_reads,), min(read_len, G), dtype=torch.int64, device=dev)
    elif length_model == "lognormal":
        n_try = int(target / (read_len * float(np.exp(sigma * sigma / 2))) * 1.2) + 4096
        L = torch.empty(n_try, device=dev, dtype=torch.float64).log_normal_(float(np.log(read_len)), sigma, generator=g)
        L = L.clamp(min_len, min(max_len, G)).to(torch.int64)
        cs = torch.cumsum(L, 0)
        n_reads = int(torch.searchsorted(cs, torch.tensor([target], device=dev, dtype=torch.int64))[0]) + 1
        n_reads = min(n_reads, n_try)
        src_len = L[:n_reads]
    elif length_model == "normal":  # HiFi-like: N(read_len, 0.1 read_len)
        n_try = int(target / read_len * 1.1) + 4096
        L = torch.empty(n_try, device=dev, dtype=torch.float64).normal_(float(read_len), 0.1 * read_len, generator=g)
        L = L.clamp(min_len, min(max_len, G)).to(torch.int64)
        cs = torch.cumsum(L, 0)
        n_reads = int(torch.searchsorted(cs, torch.tensor([target], device=dev, dtype=torch.int64))[0]) + 1
        n_reads = min(n_reads, n_try)
        src_len = L[:n_reads]
    else:
        raise ValueError(length_model)
    start = (torch.rand(n_reads, device=dev, dtype=torch.float64, generator=g) * (G - src_len + 1).to(torch.float64)).to(torch.int64)
    strand = torch.randint(0, 2, (n_reads,), dtype=torch.uint8, device=dev, generator=g)
    packed_chunks, lengths_all, woff_chunks = [], [], []
    word_total = 0
    cum = torch.cumsum(src_len, 0)
    c0 = 0
    while c0 < n_reads:
        base0 = int(cum[c0 - 1]) if c0 else 0
        c1 = int(torch.searchsorted(cum, torch.tensor([base0 + chunk_bases], device=dev, dtype=torch.int64))[0])
        c1 = max(c0 + 1, min(n_reads, c1))
        sl = src_len[c0:c1]
        m = c1 - c0
        off = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        torch.cumsum(sl, 0, out=off[1:])
        tot = int(off[-1])
        read_of = torch.repeat_interleave(torch.arange(m, device=dev), sl)
        within = torch.arange(tot, device=dev) - off[read_of]
        fwd = strand[c0:c1][read_of] == 0
        st = start[c0:c1][read_of]
        gpos = torch.where(fwd, st + within, st + sl[read_of] - 1 - within)
        base = genome[gpos]
        base = torch.where(fwd, base, 3 - base)
        u = torch.rand(tot, device=dev, generator=g)
        is_del = u < dele
        is_sub = (~is_del) & (u < dele + sub)
        base = torch.where(is_sub, (base + torch.randint(1, 4, (tot,), dtype=torch.uint8, device=dev, generator=g)) & 3, base)
        n_ins = (torch.rand(tot, device=dev, generator=g) < ins).to(torch.int64)
        emit = (~is_del).to(torch.int64) + n_ins
        out_read = torch.repeat_interleave(read_of, emit)
        out_codes = torch.repeat_interleave(base, emit)
        out_off = torch.cumsum(emit, 0)
        ins_slot = out_off[n_ins > 0] - 1
        out_codes[ins_slot] = torch.randint(0, 4, (int(ins_slot.shape[0]),), dtype=torch.uint8, device=dev, generator=g)
        lengths = torch.bincount(out_read, minlength=m)
        words, woff = pack_torch(out_codes, lengths)
        packed_chunks.append(words.cpu().numpy().view(np.uint64))
        lengths_all.append(lengths.cpu().numpy().astype(np.uint32))
        woff_chunks.append(woff[1:].cpu().numpy() + word_total)
        word_total += int(woff[-1])
        c0 = c1
    packed = np.concatenate(packed_chunks + [np.zeros(1, dtype=np.uint64)])
    lengths = np.concatenate(lengths_all)
    woffs = np.concatenate([np.zeros(1, dtype=np.int64)] + woff_chunks).astype(np.uint64)
    rs = ReadSet(packed, woffs, lengths, np.arange(n_reads, dtype=np.uint32))
    truth = dict(start=start.cpu().numpy(), src_len=src_len.cpu().numpy(), strand=strand.cpu().numpy())
    return rs, truth
