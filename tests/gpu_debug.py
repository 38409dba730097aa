"""Stage-by-stage GPU-vs-oracle debug run (not a pytest file): python tests/gpu_debug.py [--big]

Prints the first mismatches of every stage so that one gpurun call tells where parity breaks.
"""
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import oracle  # noqa: E402
from raven_amd import hip, seqio, synth  # noqa: E402
from oracle import seqio_oracle
from tests import parity_util as pu  # noqa: E402


def stage(name, fn):
    t = time.time()
    try:
        errs = fn()
        if isinstance(errs, tuple):
            errs = errs[0]
        status = "OK" if not errs else "MISMATCH"
        print("[%s] %s (%.2fs)" % (status, name, time.time() - t), flush=True)
        for e in errs[:6]:
            print("    ", e, flush=True)
        return not errs
    except Exception:
        print("[EXC] %s (%.2fs)" % (name, time.time() - t), flush=True)
        traceback.print_exc()
        return False


def run_set(tag, rs, k=15, w=5, freq=0.001, nmap=None):
    print("== %s: %d reads, %d bases, k=%d w=%d" % (tag, rs.n, rs.total_bases, k, w), flush=True)
    he = hip.Engine(k, w)
    oe = oracle.Engine(k, w)
    rd = he.upload(rs)
    n = rs.n
    nmap = n if nmap is None else min(n, nmap)
    ok = True
    ok &= stage(tag + " sketch full", lambda: pu.compare_sketch(he, oe, rd, rs, 0, min(n, 64), False))
    ok &= stage(tag + " sketch minhash", lambda: pu.compare_sketch(he, oe, rd, rs, 0, min(n, 64), True))
    for mh in (False, True):
        def idx():
            he.minimize(rd, 0, n, mh)
            oe.minimize(rs, 0, n, mh)
            return pu.compare_index(he, oe, rs, 0, n, mh)
        ok &= stage(tag + " index minhash=%s" % mh, idx)

        def flt():
            he.filter(freq)
            oe.filter(freq)
            return [] if he.occurrence == oe.occurrence else ["occurrence hip %d oracle %d" % (he.occurrence, oe.occurrence)]
        ok &= stage(tag + " filter", flt)
        ok &= stage(tag + " map (query minhash)", lambda: pu.compare_map(he, oe, rd, rs, 0, nmap, True))
        ok &= stage(tag + " map (query full, filtered)",
                    lambda: pu.compare_map(he, oe, rd, rs, 0, min(nmap, 48), False, want_filtered=True))
    for kw in (dict(use_minhash=False), dict(use_minhash=True),
               dict(use_minhash=False, index_batch_bases=rs.total_bases // 3 + 1, flush_bases=rs.total_bases // 7 + 1),
               dict(use_minhash=False, kmax=8)):
        he2 = hip.Engine(k, w)
        oe2 = oracle.Engine(k, w)
        rd2 = he2.upload(rs)
        ok &= stage(tag + " pass1 %s" % kw, lambda: pu.compare_pass1(he2, oe2, rd2, rs, freq=freq, **kw))
        print("     stage ms:", {a: round(b[0], 3) for a, b in he2.stage_ms().items()}, he2.counters(), flush=True)
    return ok


def main():
    print("devices:", hip.device_count(), flush=True)
    here = os.path.dirname(os.path.abspath(__file__))
    ok = True
    lam = seqio_oracle.load_reads(os.path.join(here, "golden", "ERA476754.fastq.gz"))
    ok &= run_set("lambda", lam)
    g = synth.make_genome(200_000, seed=11)
    rs, _ = synth.make_reads(g, 20, 8000, seed=12)
    ok &= run_set("synth200k", rs)
    rs2, _ = synth.make_reads(g, 12, 9000, length_model="lognormal", seed=13, sub=0.01, ins=0.005, dele=0.005)
    ok &= run_set("synth-lowerr-k19", rs2, k=19, w=7, nmap=64)
    print("ALL OK" if ok else "SOME FAILED", flush=True)
    if "--big" in sys.argv:
        g = synth.make_genome(5_000_000)
        t = time.time()
        rs, _ = synth.make_reads(g, 30, 10000)
        print("generated C2 in %.1fs: %d reads %d bases" % (time.time() - t, rs.n, rs.total_bases), flush=True)
        he = hip.Engine()
        rd = he.upload(rs)
        for it in range(3):
            he.reset_stats()
            t = time.time()
            p = he.find_overlaps_and_create_piles(rd)
            dt = time.time() - t
            print("C2 pass1 iter %d: %.3fs -> %.3f Gbase/s; overlaps kept %d" % (
                it, dt, rs.total_bases / dt / 1e9, len(p.overlaps()[0])), flush=True)
            print("   stage ms:", {a: round(b[0], 2) for a, b in he.stage_ms().items()}, flush=True)
            print("   counters:", he.counters(), flush=True)
            p.close()


if __name__ == "__main__":
    main()
