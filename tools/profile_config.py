#!/usr/bin/env python3
"""Wall time and per-kernel device time of every stage of the overlap + polish path on a synthetic config:
    python tools/profile_config.py --config hifi --genome 20000000
Stages: first pass, TrimAndAnnotate, identity filter (if --identity), second pass, one polishing round."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raven_amd import hip, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["ont", "hifi", "ont10k"], default="hifi")
    ap.add_argument("--genome", type=int, default=20_000_000)
    ap.add_argument("--identity", type=float, default=None)
    ap.add_argument("--skip", default="")
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    cov, rl, model, errs = {"ont": (30, 9000, "lognormal", (0.04, 0.03, 0.03)), "ont10k": (30, 10000, "fixed", (0.04, 0.03, 0.03)),
                            "hifi": (40, 15000, "normal", (0.001, 0.002, 0.002))}[args.config]
    identity = args.identity if args.identity is not None else (0.95 if args.config == "hifi" else 0.0)
    t = time.time()
    g = synth.make_genome_torch(args.genome, seed=11, device=dev)
    rs, truth = synth.make_reads_torch(g, cov, rl, length_model=model, sub=errs[0], ins=errs[1], dele=errs[2], seed=12)
    out = {"config": args.config, "genome": args.genome, "reads": rs.n, "bases": rs.total_bases, "gen_s": round(time.time() - t, 2)}
    eng = hip.Engine(15, 5)
    rd = eng.upload(rs)
    eng.set_timing(False)
    eng.set_kernel_timing(True)

    def stage(name, fn):
        eng.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        km = sorted(eng.kernel_ms().items(), key=lambda x: -x[1][0])
        out[name] = {"wall_s": round(dt, 3), "kernels_ms": {k: (round(v[0], 1), v[1]) for k, v in km[:10] if v[1]},
                     "counters": eng.counters()}
        print(name, json.dumps(out[name]), flush=True)
        return r

    p = stage("pass1_warm", lambda: eng.find_overlaps_and_create_piles(rd))
    p.close()
    p = stage("pass1", lambda: eng.find_overlaps_and_create_piles(rd))
    ovl, off = p.overlaps()
    begin, end, median, invalid = stage("trim", lambda: p.trim_and_annotate(4))
    p.close()
    begin, end = (begin.astype(np.uint32) << 4), (end.astype(np.uint32) << 4)
    out["overlaps_kept"] = int(ovl.shape[0])
    if identity and "identity" not in args.skip:
        kept, koff = stage("identity_filter", lambda: eng.filter_overlaps_by_identity(rd, ovl, off, begin, end, invalid, identity))
        out["identity_kept"] = int(kept.shape[0])
    if "pass2" not in args.skip:
        # containment is host graph logic in raven; emulate its effect (most reads contained) with a random 70 %
        rng = np.random.default_rng(1)
        inv2 = (invalid | (rng.random(rs.n) < 0.7)).astype(np.uint8)
        r2 = stage("pass2", lambda: eng.find_overlaps_and_repetitive_regions(rd, begin, end, inv2, identity=identity))
        out["pass2_overlaps"] = int(r2["overlaps"].shape[0])
    if "polish" not in args.skip:
        n = args.genome
        bounds = np.linspace(0, n, max(1, n // 5_000_000) + 1).astype(np.int64)
        drafts = [synth.mutate_torch(g[int(bounds[i]):int(bounds[i + 1])], 0.01, 0.008, 0.008, seed=50 + i).cpu().numpy()
                  for i in range(len(bounds) - 1)]
        td = eng.upload_codes(drafts)
        cons, ratio, st = stage("polish_round", lambda: eng.polish_round(td, rd))
        out["polish_stats"] = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}
        cons, ratio, st = stage("polish_round_2nd_call", lambda: eng.polish_round(td, rd))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
