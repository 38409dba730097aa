#!/usr/bin/env python3
"""bench.py — read Gbase/s through raven's overlap hot path (FindOverlapsAndCreatePiles) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): synthetic 5 Mb genome, 30x ONT-like 10 kb reads (10 % error),
k=15 w=5, -p 0 (no polishing rounds): one "step" = one full raven::FindOverlapsAndCreatePiles pass over the
whole read set — sketch, index (sort), filter, map/chain of every read, merge, Pile::AddLayers, top-kMax
truncation — with the packed reads already resident in HBM and results left in HBM.
Multi-GPU: every rank owns an independent shard (its own genome + reads), no data-path collective ->
"scaling": "weak"; value = bases processed by all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 (metric contract + "roofline" + "cpu_baseline").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raven_amd import dist as rdist  # noqa: E402
from raven_amd import hip, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json,
    made by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this same bench command):
    2 x FETCH_SIZE KiB (gfx950: FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md §HBM) + WRITE_SIZE KiB."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        e = t["kernels"].get(kernel)
        return int(e["hbm_bytes_per_launch"]) if e else None
    except (OSError, ValueError, KeyError):
        return None


def algorithmic_bytes(site, c, val_bytes):
    """ALGORITHMIC bytes of ONE launch of kernel `site`, from the exact engine counters of
    one step: N index bases, Mi index minimizers, U keys, Mq query minimizers, H matches, O overlaps."""
    N, Mi, U, Mq, H, O = (c["index_bases"], c["index_minimizers"], c["index_keys"], c["query_minimizers"],
                          c["matches"], c["overlaps"])
    rec = val_bytes + 8  # one (value, origin) record
    table = {
        "sketch_count": N / 4.0,
        "sketch_write": N / 4.0 + rec * Mi,
        "rs_upsweep": val_bytes * Mi,
        "rs_downsweep": 2.0 * rec * Mi,          # read + write every (value, origin) once
        "heads": (val_bytes + 1) * Mi,
        "unique": (val_bytes + 1 + 4) * Mi + (val_bytes + 4) * U,
        "scan": None,
        "table": val_bytes * U + 4.0 * (1 << 26),
        "match_count": rec * Mq + 8.0 * H + 12.0 * Mq,
        "match_emit": 8.0 * Mq + 8.0 * H + 16.0 * H,
        "seg_sort_group": 2.0 * 16.0 * H,
        "seg_sort_pos": 2.0 * 16.0 * H,
        "chain": 16.0 * H + 32.0 * O,
        "minhash_select": val_bytes * Mi + Mi,
    }
    return table.get(site)


def cpu_baseline(args, cores):
    """Oracle (CPU restatement, 'port') timed on a bounded sample of the same workload."""
    from oracle import oracle
    g = synth.make_genome(args.cpu_sample_genome, seed=0xC0FFEE)
    rs, _ = synth.make_reads(g, args.coverage, args.read_len, seed=0xC0FFEF)
    t = time.time()
    r1 = oracle.Engine(args.k, args.w).find_overlaps_and_create_piles(rs, threads=1) if args.cpu_single else None
    t1 = time.time() - t
    t = time.time()
    oracle.Engine(args.k, args.w).find_overlaps_and_create_piles(rs, threads=cores)
    tc = time.time() - t
    sample = "%d reads / %d bases of the same generator (%.2f Mb genome, %gx), oracle FindOverlapsAndCreatePiles, %d threads: %.2f s" % (
        rs.n, rs.total_bases, args.cpu_sample_genome / 1e6, args.coverage, cores, tc)
    if r1 is not None:
        sample += "; 1 thread: %.2f s = %.4f Gbase/s" % (t1, rs.total_bases / t1 / 1e9)
    return {"value": rs.total_bases / tc / 1e9, "unit": "Gbase/s", "cores": cores, "kind": "port", "sample": sample}


def cpu_baseline_polish():
    """The CPU restatement of one racon round (oracle.polish_round, 1 thread: whole-overlap NW path + spoa-style POA)
    on a bounded sample of the same generator: 20 kb draft, 30x, 5 kb reads."""
    from oracle import oracle
    from raven_amd import seqio
    g = synth.make_genome(20_000, seed=0x5EED0003)
    rs, _ = synth.make_reads(g, 30, 5000, seed=0x5EED0004)
    targets = seqio.pack_reads([synth.make_draft(g, seed=0x5EED0005)])
    t0 = time.perf_counter()
    oracle.polish_round(targets, rs)
    dt = time.perf_counter() - t0
    return {"value": round(rs.total_bases / dt / 1e9, 6), "unit": "Gbase/s per round", "cores": 1, "kind": "port",
            "sample": "%d reads / %d bases on a 20 kb draft (5 kb reads: the checker aligns every read with a full "
                      "NW matrix), oracle.polish_round: %.1f s" % (rs.n, rs.total_bases, dt)}


def bench_sharded(args, rank, world, local_rank, dist, barrier):
    """Strong-scaling leg: every rank generates the SAME genome/reads, owns a slice of the piles, and the pass runs
    through raven_amd/sharded.py (all-to-all over RCCL on torch CUDA tensors; nothing crosses PCIe between stages)."""
    from raven_amd import sharded
    genome_seed, reads_seed = rdist.shard_seeds(0)
    genome = synth.make_genome(args.genome, seed=genome_seed)
    rs, _ = synth.make_reads(genome, args.coverage, args.read_len, seed=reads_seed)
    eng = hip.Engine(args.k, args.w, device=local_rank)
    eng.set_timing(False)
    import torch
    comm = sharded.DeviceComm(dist, device="cuda")
    dev = torch.device("cuda", local_rank)
    res = None
    for _ in range(args.warmup):
        res = sharded.find_overlaps_and_create_piles_sharded_dev(eng, rs, comm, dev, freq=args.freq, kmax=args.kmax)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = sharded.find_overlaps_and_create_piles_sharded_dev(eng, rs, comm, dev, freq=args.freq, kmax=args.kmax)
    barrier()
    dt = time.perf_counter() - t0
    dt, _ = rdist.aggregate(dt, 0.0, dist, device="cuda")
    if rank == 0:
        steps = max(args.steps, 1)
        print(json.dumps({
            "metric": "read Gbase/s through overlap+polish", "value": round(rs.total_bases * steps / dt / 1e9, 4),
            "unit": "Gbase/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1] as ONE genome sharded over the ranks: %.1f Mb, %gx, %d bp "
                                   "reads, -p 0" % (args.genome / 1e6, args.coverage, args.read_len),
                       "parallelism": "reads by pile, minimizers by hash class; all-to-all x3 on CUDA tensors (RCCL)",
                       "rank0": {k: res[k] for k in ("lo", "hi", "occurrence", "stats")}},
            "roofline": None, "cpu_baseline": None}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--w", type=int, default=5)
    ap.add_argument("--freq", type=float, default=0.001)
    ap.add_argument("--kmax", type=int, default=32)
    ap.add_argument("--cpu-sample-genome", type=int, default=1_000_000)
    ap.add_argument("--cpu-single", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--sharded", action="store_true", help="ONE genome sharded across the ranks (SURVEY 8(e): reads by "
                    "pile, minimizers by hash class, three all-to-all exchanges) instead of one independent shard per "
                    "GPU; strong scaling, overlap pass only")
    ap.add_argument("--no-polish", action="store_true", help="skip the configs[2] polishing leg (reported beside, "
                    "never part of `value`)")
    ap.add_argument("--polish-rounds", type=int, default=2)
    args = ap.parse_args()

    rank, world, local_rank = rdist.env_rank()

    import torch  # plumbing only: device selection, barrier, max-over-ranks
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.sharded:
        return bench_sharded(args, rank, world, local_rank, dist, barrier)

    # ---- synthetic shard of this rank (seeded; rank-dependent so shards are independent) ----
    t0 = time.time()
    genome_seed, reads_seed = rdist.shard_seeds(rank)
    genome = synth.make_genome(args.genome, seed=genome_seed)
    rs, _ = synth.make_reads(genome, args.coverage, args.read_len, seed=reads_seed)
    t_gen = time.time() - t0

    eng = hip.Engine(args.k, args.w, device=local_rank)
    t0 = time.time()
    reads = eng.upload(rs)  # H2D once; resident for every step
    t_h2d = time.time() - t0
    eng.set_timing(False)  # no per-stage host syncs inside the timed region
    eng.set_kernel_timing(not args.no_kernel_timing)

    def step():
        p = eng.find_overlaps_and_create_piles(reads, freq=args.freq, kmax=args.kmax)  # synchronous: returns when done
        p.close()

    for _ in range(args.warmup):
        step()
    eng.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0

    dt, total_bases = rdist.aggregate(dt, float(rs.total_bases), dist, device="cuda")

    counters_raw = eng.counters()
    kms_raw = eng.kernel_ms() if not args.no_kernel_timing else {}
    eng.reset_stats()

    # ---- configs[2] leg: -p 2, racon-style polishing rounds of this shard's draft assembly with the same reads ----
    polish = None
    if not args.no_polish and args.polish_rounds > 0:
        from raven_amd import seqio
        peng = eng if (args.k, args.w) == (15, 5) else hip.Engine(15, 5, device=local_rank)  # racon maps with (15, 5)
        preads = reads if peng is eng else peng.upload(rs)
        draft = synth.make_draft(genome, seed=genome_seed + 7)
        peng.polish_round(peng.upload(seqio.pack_reads([draft[:100_000]])), preads)  # warm-up (allocations)
        peng.reset_stats()
        cur = seqio.pack_reads([draft])
        barrier()
        tp0 = time.perf_counter()
        last = None
        n_windows = 0
        for _ in range(args.polish_rounds):
            cons, ratio, last = peng.polish_round(peng.upload(cur), preads)
            n_windows += last["n_windows"]
            cur = seqio.pack_reads([cons[0]])
        barrier()
        dtp = time.perf_counter() - tp0
        dtp, _ = rdist.aggregate(dtp, float(rs.total_bases), dist, device="cuda")
        polish = {
            "workload": "BASELINE.json configs[2]: same genome/reads, -p %d: draft = genome with 2.6%% iid errors, "
                        "racon-style rounds (map reads to the draft, 500-bp windows, POA consensus m/n/g = 3/-5/-4)"
                        % args.polish_rounds,
            "rounds": args.polish_rounds, "s_per_round": round(dtp / args.polish_rounds, 4),
            "read_gbase_per_s_per_round": round(total_bases / (dtp / args.polish_rounds) / 1e9, 4),
            "windows_per_s": round(n_windows * world / dtp, 1),
            "overlap_plus_polish_gbase_per_s": round(total_bases / (dt / max(args.steps, 1) + dtp) / 1e9, 4),
            "last_round": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in last.items()},
            "polished_ratio": round(float(ratio[0]), 4),
            "kernels_ms_per_round": {k: round(v[0] / args.polish_rounds, 3) for k, v in
                                     sorted(peng.kernel_ms().items(), key=lambda x: -x[1][0])[:6] if v[1]}
            if not args.no_kernel_timing else None,
            "windows_rerun_wide_band": peng.poa_wide_windows(), "windows_rerun_full_matrix": peng.poa_fallback_windows(),
        }

    if rank == 0:
        steps = max(args.steps, 1)
        counters = {k: v // steps for k, v in counters_raw.items()}
        kms = kms_raw
        val_bytes = 4 if 2 * eng.k < 32 else 8
        roofline = None
        kernels = {}
        if kms:
            tot = sum(v[0] for v in kms.values())
            for name, (ms, la) in sorted(kms.items(), key=lambda x: -x[1][0]):
                if la:
                    kernels[name] = {"ms_per_step": round(ms / steps, 4), "launches_per_step": la / steps,
                                     "avg_launch_ms": round(ms / la, 5)}
            dom = None
            for name in kernels:  # dominant kernel with a defined algorithmic byte count
                if algorithmic_bytes(name, counters, val_bytes):
                    dom = name
                    break
            if dom:
                b = algorithmic_bytes(dom, counters, val_bytes)
                avg_s = kms[dom][0] / kms[dom][1] / 1e3
                achieved = b / avg_s / 1e9
                roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom),
                            "algorithmic_bytes_per_launch": int(b), "avg_launch_ms": round(avg_s * 1e3, 5),
                            "kernel_ms_share": round(kms[dom][0] / tot, 3) if tot else None}
        out = {
            "metric": "read Gbase/s through overlap+polish",
            "value": round(rdist.throughput(dt, total_bases, args.steps) / 1e9, 4),
            "unit": "Gbase/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32" if val_bytes == 4 else "u64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[1]: synthetic %.1f Mb genome, %gx ONT-like %d bp reads "
                            "(4%% sub, 3%% ins, 3%% del), k=%d w=%d, -p 0 (FindOverlapsAndCreatePiles only, "
                            "no polishing rounds), 1 shard per GPU" % (args.genome / 1e6, args.coverage,
                                                                       args.read_len, args.k, args.w),
                "reads_per_gpu": rs.n, "bases_per_gpu": rs.total_bases, "freq": args.freq, "kmax": args.kmax,
                "parallelism": "independent shard per GPU (no data-path collective)",
            },
            "overlaps_per_s": round(counters["overlaps"] * world * args.steps / dt, 1),
            "counters_per_step": counters,
            "roofline": roofline,
            "kernels": kernels,
            "polish": polish,
            "host": {"gen_s": round(t_gen, 2), "h2d_s": round(t_h2d, 3),
                     "h2d_inclusive_gbase_s": round(rs.total_bases / (dt / steps + t_h2d) / 1e9, 4)},
        }
        if not args.no_cpu_baseline and world == 1:
            cores = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_baseline(args, cores)
            if polish is not None:
                polish["cpu_baseline"] = cpu_baseline_polish()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
