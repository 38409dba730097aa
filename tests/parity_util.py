"""Stage-by-stage HIP-vs-oracle comparison helpers shared by the GPU parity tests and the debug script.

Every function returns a list of human-readable mismatch strings (empty == parity).
"""
from __future__ import annotations

import numpy as np


def first_diff(a: np.ndarray, b: np.ndarray):
    n = min(a.shape[0], b.shape[0])
    if n:
        neq = np.nonzero(a[:n] != b[:n])[0]
        if neq.size:
            return int(neq[0])
    if a.shape[0] != b.shape[0]:
        return n
    return None


def compare_sketch(hip_eng, orc_eng, reads_dev, rs, first, last, minhash):
    errs = []
    v, o, off = hip_eng.sketch(reads_dev, first, last, minhash)
    for i in range(first, last):
        ov, oo = orc_eng.sketch(rs, i, minhash)
        hv = v[off[i - first]: off[i - first + 1]]
        ho = o[off[i - first]: off[i - first + 1]]
        d = first_diff(hv, ov)
        if d is None:
            d = first_diff(ho, oo)
        if d is not None:
            errs.append("sketch read %d (len %d, minhash=%s): first diff at %d; hip n=%d orc n=%d; hip=%s orc=%s" % (
                i, rs.lengths[i], minhash, d, hv.shape[0], ov.shape[0],
                [(int(x), int(y) & 0xFFFFFFFF) for x, y in zip(hv[d:d + 3], ho[d:d + 3])],
                [(int(x), int(y) & 0xFFFFFFFF) for x, y in zip(ov[d:d + 3], oo[d:d + 3])]))
            if len(errs) > 5:
                break
    return errs


def compare_index(hip_eng, orc_eng, rs, first, last, minhash, sample=2000, seed=1):
    """hip sorted index vs oracle Find() on a sample of keys + global invariants."""
    errs = []
    v, o, u = hip_eng.index_content()
    if v.shape[0] > 1 and np.any(v[1:] < v[:-1]):
        errs.append("index values not sorted")
    c = orc_eng.counters()
    rng = np.random.default_rng(seed)
    if v.shape[0]:
        heads = np.nonzero(np.concatenate([[True], v[1:] != v[:-1]]))[0]
        if heads.shape[0] != u:
            errs.append("distinct keys: hip reports %d, array has %d" % (u, heads.shape[0]))
        ends = np.concatenate([heads[1:], [v.shape[0]]])
        pick = rng.choice(heads.shape[0], size=min(sample, heads.shape[0]), replace=False)
        for j in pick:
            val = int(v[heads[j]])
            ho = o[heads[j]: ends[j]]
            oo, n = orc_eng.find(val)
            if n != ho.shape[0] or first_diff(ho, oo[:n]) is not None:
                errs.append("index key %d: hip origins %s vs oracle %s" % (val, ho[:4], oo[:4]))
                if len(errs) > 5:
                    break
    return errs


def compare_map(hip_eng, orc_eng, reads_dev, rs, first, last, minhash, avoid_equal=True, avoid_symmetric=True,
                want_filtered=False):
    errs = []
    res = hip_eng.map_batch(reads_dev, first, last, avoid_equal, avoid_symmetric, minhash, want_filtered)
    ovl, off = res["overlaps"], res["read_offsets"]
    n_ovl = 0
    for i in range(first, last):
        r = orc_eng.map(rs, i, avoid_equal, avoid_symmetric, minhash)
        ho = ovl[off[i - first]: off[i - first + 1]]
        oo = r["overlaps"]
        n_ovl += oo.shape[0]
        if ho.shape[0] != oo.shape[0] or (ho.shape[0] and not np.array_equal(ho, oo)):
            d = first_diff(ho, oo)
            errs.append("map read %d: hip %d overlaps, oracle %d; first diff %s: hip=%s orc=%s" % (
                i, ho.shape[0], oo.shape[0], d, ho[d:d + 1] if d is not None else None,
                oo[d:d + 1] if d is not None else None))
        if want_filtered:
            hf = res["filtered"][res["filtered_offsets"][i - first]: res["filtered_offsets"][i - first + 1]]
            if not np.array_equal(hf, r["filtered"]):
                errs.append("map read %d: filtered positions differ (hip %d, oracle %d)" % (
                    i, hf.shape[0], r["filtered"].shape[0]))
        if len(errs) > 5:
            break
    return errs, n_ovl


def compare_pass1(hip_eng, orc_eng, reads_dev, rs, **kw):
    errs = []
    p = hip_eng.find_overlaps_and_create_piles(reads_dev, **kw)
    ref = orc_eng.find_overlaps_and_create_piles(rs, **kw)
    data, poff = p.piles()
    ovl, ooff = p.overlaps()
    if hip_eng.occurrence != ref["occurrence"]:
        errs.append("occurrence: hip %d oracle %d" % (hip_eng.occurrence, ref["occurrence"]))
    if not np.array_equal(poff, ref["pile_offsets"]):
        errs.append("pile offsets differ")
    elif not np.array_equal(data, ref["pile_data"]):
        d = first_diff(data, ref["pile_data"])
        pile = int(np.searchsorted(poff, d, side="right") - 1)
        errs.append("pile data differ: first at word %d (pile %d, cell %d): hip %d oracle %d; total diff cells %d" % (
            d, pile, d - int(poff[pile]), data[d], ref["pile_data"][d], int((data != ref["pile_data"]).sum())))
    if not np.array_equal(ooff.astype(np.uint64), ref["overlap_offsets"]):
        d = first_diff(ooff.astype(np.uint64), ref["overlap_offsets"])
        errs.append("overlap offsets differ first at pile %s: hip %s oracle %s" % (
            d, ooff[d:d + 2], ref["overlap_offsets"][d:d + 2]))
    elif not np.array_equal(ovl, ref["overlaps"]):
        d = first_diff(ovl, ref["overlaps"])
        pile = int(np.searchsorted(ooff, d, side="right") - 1)
        errs.append("overlaps differ first at %d (pile %d): hip %s oracle %s; total diff %d" % (
            d, pile, ovl[d], ref["overlaps"][d], int((ovl != ref["overlaps"]).sum())))
    hc, oc = hip_eng.counters(), ref["counters"]
    for key in ("index_bases", "index_minimizers", "index_keys", "query_bases", "query_minimizers", "matches",
                "overlaps"):
        if hc[key] != oc[key]:
            errs.append("counter %s: hip %d oracle %d" % (key, hc[key], oc[key]))
    p.close()
    return errs, ref
