"""Definitional (brute-force) numpy/python restatements used to pin the C++ oracle:
minimizer sketch straight from the published definition, naive pile coverage, naive LIS validity."""
from __future__ import annotations

import numpy as np

U64 = np.uint64


def _hash(key: np.ndarray, mask: int) -> np.ndarray:
    m = U64(mask)
    with np.errstate(over="ignore"):
        key = ((~key) + (key << U64(21))) & m
        key = key ^ (key >> U64(24))
        key = ((key + (key << U64(3))) + (key << U64(8))) & m
        key = key ^ (key >> U64(14))
        key = ((key + (key << U64(2))) + (key << U64(4))) & m
        key = key ^ (key >> U64(28))
        key = (key + (key << U64(31))) & m
    return key


def brute_sketch(codes: np.ndarray, k: int, w: int, read_id: int = 0, minhash: bool = False):
    """(values, origins) from the definition: canonical k-mer hash (palindromes skipped), a position is
    a minimizer iff it attains the minimum of some full window of w consecutive k-mer positions."""
    n = codes.shape[0]
    if n < k:
        return np.zeros(0, U64), np.zeros(0, U64)
    P = n - k + 1
    c = codes.astype(U64)
    fwd = np.zeros(P, U64)
    rev = np.zeros(P, U64)
    for j in range(k):
        fwd |= c[j:j + P] << U64(2 * (k - 1 - j))
        rev |= (U64(3) - c[j:j + P]) << U64(2 * j)
    mask = (1 << (2 * k)) - 1
    valid = fwd != rev
    strand = (fwd > rev)
    h = _hash(np.where(strand, rev, fwd), mask)
    INF = U64(0xFFFFFFFFFFFFFFFF)
    h = np.where(valid, h, INF)
    sel = np.zeros(P, bool)
    if P >= w:
        nwin = P - w + 1
        win = np.lib.stride_tricks.sliding_window_view(h, w)  # [nwin, w]
        mins = win.min(axis=1)
        hit = (win == mins[:, None]) & (mins[:, None] != INF)
        for q in range(w):
            sel[q:q + nwin] |= hit[:, q]
    pos = np.nonzero(sel)[0].astype(U64)
    vals = h[sel]
    org = (U64(read_id) << U64(32)) | (pos << U64(1)) | strand[sel].astype(U64)
    if minhash:
        keep = min(vals.shape[0], n // k)
        order = np.argsort(vals, kind="stable")[:keep]
        order.sort()
        vals, org = vals[order], org[order]
    return vals, org


def naive_add_layers(data: np.ndarray, pile_id: int, overlaps: np.ndarray) -> np.ndarray:
    """Per-cell counting restatement of Pile::AddLayers for sane overlaps (end event after begin event)."""
    out = data.astype(np.int64).copy()
    cov = np.zeros(data.shape[0] + 2, np.int64)
    for o in overlaps:
        if o["lhs_id"] == pile_id:
            b, e = int(o["lhs_begin"]), int(o["lhs_end"])
        elif o["rhs_id"] == pile_id:
            b, e = int(o["rhs_begin"]), int(o["rhs_end"])
        else:
            continue
        cov[(b >> 4) + 1] += 1
        cov[(e >> 4) - 1] -= 1
    cov = np.cumsum(cov)[:data.shape[0]]
    return np.minimum(out + cov, 65535).astype(np.uint16)


def edit_distance(a: bytes, b: bytes) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j - 1] + (ca != cb), prev[j] + 1, cur[j - 1] + 1))
        prev = cur
    return prev[-1]
