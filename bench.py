#!/usr/bin/env python3
"""bench.py — read Gbase/s through raven's overlap + polish hot path (FindOverlapsAndCreatePiles + Polish) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload = the configuration BASELINE.json's metric is quoted on (configs[3] on ONE GPU at N = 1, sharded over the ranks
at N > 1): synthetic 100 Mb genome, 30x ONT-length reads (log-normal, median 9 kb, 10 % errors), k = 15 w = 5, -p 2.
One "step" = the whole hot path over the whole read set:
    raven::FindOverlapsAndCreatePiles   sketch, index (sort), filter, map/chain of every read in flush windows of 2^30
                                        bases, merge, Pile::AddLayers, top-kMax truncation             (construct.cc:14-121)
  + 2 x racon::Polisher::Polish         map reads to the draft contigs, best overlap, alignment path + breakpoints,
                                        window layers, POA consensus, stitch; the consensus of round 1 is the target of
                                        round 2                                                         (polish.cc:43-74)
with the packed reads already resident in HBM when the timed region starts.  value = read bases x steps / wall time of
the timed region (barrier + synchronize on both sides, max over ranks).
--workload c2 runs BASELINE.json configs[2] instead (5 Mb, 10 kb reads, -p 2); --polish-rounds 0 gives configs[1].
N > 1: ONE genome sharded across the ranks ("scaling": "strong"): reads by pile, minimizers by hash class, three
all-to-alls + one all-reduce per flush window (raven_amd/sharded.py), polishing windows sharded by range.

Prints ONE JSON line on rank 0 (metric contract + "roofline" + "cpu_baseline").
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raven_amd import dist as rdist  # noqa: E402
from raven_amd import hip, seqio, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
# integer VALU peak: 256 CU x 4 SIMD x 32 lanes... counted as 64-lane wave instructions:
# 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction = 1.23e12 wave-instructions/s = 78.6e12 lane-ops/s
VALU_PEAK_LANE_OPS = 78.6e12
# a linear-gap POA cell needs at least: one add + one max per candidate (diagonal, vertical per predecessor,
# horizontal) ~ 6 lane-ops; the graded "peak" cell rate is the lane-op peak / 6
POA_MIN_OPS_PER_CELL = 6.0
NW_MIN_OPS_PER_CELL = 50.0 / 64.0  # Myers block update + match mask: ~25 64-bit operations per 64 cells


def kernel_source_hash():
    """sha1 over the kernels' sources (raven_amd/csrc/*.hip, *.h, sorted): what a traffic profile is valid for."""
    import glob
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "raven_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


_TRAFFIC = None


def pmc_traffic_file():
    """This round's rocprofv3 --pmc result (profiles/r06_pmc_traffic.json, made by tools/pmc_traffic.py from separate
    FETCH_SIZE / WRITE_SIZE passes of this same bench command; tools/profile_round.sh stamps it with the hash of the
    kernel sources it was taken on)."""
    global _TRAFFIC
    if _TRAFFIC is None:
        _TRAFFIC = {}
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json"):  # (the newest there is; `stale` says whether it fits the sources)
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    _TRAFFIC = json.load(f)
                _TRAFFIC["file"] = "profiles/" + name
                break
            except (OSError, ValueError):
                pass
    return _TRAFFIC


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel`: 2 x FETCH_SIZE KiB (gfx950: FETCH_SIZE counts 128-B requests as 64 B,
    MI355X_MICROARCH.md, HBM) + WRITE_SIZE KiB — the guide's reading, calibrated on wide coalesced streams; None when this
    round has no profile of it."""
    e = pmc_traffic_file().get("kernels", {}).get(kernel)
    return int(e["hbm_bytes_per_launch"]) if e else None


def pmc_traffic_raw(kernel):
    """The same counters as rocprofv3 reports them: FETCH_SIZE KiB + WRITE_SIZE KiB per launch (what a kernel of 4- or
    16-byte gathers is counted at: 64 B per probe, profiles/r06_pmc_traffic.json "calibration")."""
    e = pmc_traffic_file().get("kernels", {}).get(kernel)
    return int(e["hbm_bytes_raw_per_launch"]) if e and "hbm_bytes_raw_per_launch" in e else None


def traffic_provenance():
    t = pmc_traffic_file()
    if not t.get("kernels"):
        return None
    sha = t.get("kernel_source_sha1")
    return {"file": t.get("file"), "kernel_source_sha1": sha,
            "stale": (sha != kernel_source_hash()) if sha else True,
            "calibration_raw_counter_over_known_bytes": {k: v.get("raw_over_known") for k, v in (t.get("calibration") or {}).items()
                                                         if isinstance(v, dict) and "raw_over_known" in v},
            "note": "measured by tools/profile_round.sh (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this command); "
                    "stale = the kernel sources changed since.  'traffic' = 2 x FETCH + WRITE (the guide's correction, right "
                    "for streams), 'traffic_raw' = FETCH + WRITE; the calibration says what the raw counters read on this box "
                    "for a known byte count per access pattern (tools/pmc_calibrate.hip)"}


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
        "table": val_bytes * U + 4.0 * (1 << 26),
        "match_count": rec * Mq + 8.0 * H + 12.0 * Mq,
        "match_emit": 8.0 * Mq + 8.0 * H + 16.0 * H,
        "seg_sort_group": 2.0 * 16.0 * H,
        "seg_sort_pos": 2.0 * 8.0 * H,  # (keys only since round 6)
        "chain": 16.0 * H + 32.0 * O,
        "minhash_select": val_bytes * Mi + Mi,
    }
    return table.get(site)


def cpu_baseline(args, cores):
    """The CPU restatement ('port') of the same path timed on bounded samples of the same generator, on this box's
    cores: the overlap pass multi-threaded on one sample, the polishing round as `threads` independent samples in
    parallel (the restatement of racon's round is single-threaded per call).  Combined like the metric:
    bases / (t_overlap + rounds x t_round) per base."""
    from oracle import oracle
    threads = max(1, cores)  # every core of the box
    cpu_model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    if args.cpu_sample_genome is None:  # large enough that Filter and the flush logic do something; bounded by the cores
        args.cpu_sample_genome = 20_000_000 if threads >= 96 else (8_000_000 if threads >= 32 else 2_000_000)
    g = synth.make_genome(args.cpu_sample_genome, seed=0xC0FFEE)
    rs, _ = synth.make_reads(g, args.coverage, 10000, seed=0xC0FFEF)
    t = time.time()
    oracle.Engine(args.k, args.w).find_overlaps_and_create_piles(rs, threads=threads)
    t_ovl = time.time() - t
    v_ovl = rs.total_bases / t_ovl
    sample = "overlap: %d reads / %d bases (%.1f Mb genome, %gx, 10 kb reads), oracle FindOverlapsAndCreatePiles on %d " \
             "threads: %.2f s = %.4f Gbase/s" % (rs.n, rs.total_bases, args.cpu_sample_genome / 1e6, args.coverage, threads,
                                                 t_ovl, v_ovl / 1e9)
    v = v_ovl
    if args.polish_rounds > 0:
        cases = []
        for i in range(min(threads, 128)):  # (one independent round per thread; more than 128 samples add nothing)
            gg = synth.make_genome(12_000, seed=0x5EED0003 + i)
            prs, _ = synth.make_reads(gg, 30, 3000, seed=0x5EED1004 + i)
            cases.append((seqio.pack_reads([synth.make_draft(gg, seed=0x5EED2005 + i)]), prs))
        t = time.time()
        ths = [threading.Thread(target=oracle.polish_round, args=c) for c in cases]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        t_pol = time.time() - t
        pol_bases = sum(c[1].total_bases for c in cases)
        v_pol = pol_bases / t_pol
        sample += "; polish: %d independent rounds in parallel (12 kb draft, 30x, 3 kb reads each; %d bases), " \
                  "oracle.polish_round (Ukkonen-banded scalar NW paths + scalar POA: not racon's bit-vector edlib + SIMD spoa): %.2f s = %.5f Gbase/s per round" % (len(cases), pol_bases, t_pol, v_pol / 1e9)
        v = 1.0 / (1.0 / v_ovl + args.polish_rounds / v_pol)
    return {"value": round(v / 1e9, 6), "unit": "Gbase/s", "cores": threads, "cpu_count": os.cpu_count(), "cpu_model": cpu_model,
            "kind": "port", "sample": sample}


def make_workload(args, device):
    """Seeded synthetic genome, reads and draft contigs of the workload, generated on the GPU (torch as the random /
    scatter engine), handed over as host arrays like any caller's data."""
    import torch
    t0 = time.time()
    dev = torch.device("cuda", device)
    genome = synth.make_genome_torch(args.genome, seed=0x5EED0001, device=dev)
    rs, _ = synth.make_reads_torch(genome, args.coverage, args.read_len, length_model=args.length_model, sub=args.errors[0],
                                   ins=args.errors[1], dele=args.errors[2], seed=0x5EED0002)
    drafts = []
    if args.polish_rounds > 0:
        n_contigs = max(1, (args.genome + args.contig - 1) // args.contig)
        bounds = np.linspace(0, args.genome, n_contigs + 1).astype(np.int64)
        for i in range(n_contigs):  # what raven's layout hands to racon: the truth with ~2.6 % errors, in contigs
            d = synth.mutate_torch(genome[int(bounds[i]):int(bounds[i + 1])], 0.01, 0.008, 0.008, seed=0x5EED0007 + i)
            drafts.append(d.cpu().numpy())
    del genome
    torch.cuda.empty_cache()
    return rs, drafts, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    # two warm-up steps: the engine's scratch grows to its steady size over the first passes (the sketch and the index
    # exchange buffers by ownership between the overlap pass and the polishing rounds, so a buffer can still be too
    # small for its new role in the third pass: 8.8 GB re-allocated inside `minimize`, + 0.38 s — DESIGN.md section 5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["c4", "c2", "c5"], default="c4",
                    help="c4: 100 Mb, 30x ONT-length reads (configs[3]; the metric's config).  c2: 5 Mb, 10 kb reads "
                         "(configs[2]).  c5: 100 Mb, 40x HiFi 15 kb reads, --identity 0.95 (configs[4]'s workload on one GPU: "
                         "first pass, trimming, identity filter, second pass, two rounds)")
    ap.add_argument("--identity", type=float, default=None)
    ap.add_argument("--genome", type=int, default=None)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--length-model", default=None)
    ap.add_argument("--contig", type=int, default=5_000_000, help="draft contig (unitig) length for the polishing rounds")
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--w", type=int, default=5)
    ap.add_argument("--freq", type=float, default=0.001)
    ap.add_argument("--kmax", type=int, default=32)
    ap.add_argument("--polish-rounds", type=int, default=2)
    ap.add_argument("--cpu-sample-genome", type=int, default=None, help="genome of the CPU baseline's overlap sample "
                    "(default: 20 Mb on >= 96 cores, 8 Mb on >= 32, else 2 Mb)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--load-bases", type=int, default=150_000_000,
                    help="bases of the gz FASTQ the input path (rvn_reads_load) is timed on, outside the timed region (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--engine-option", action="append", default=[], metavar="NAME=VALUE",
                    help="rvn_engine_set_option before the run (tuning experiments; repeatable)")
    ap.add_argument("--targets-through-host", action="store_true", help="round r + 1's targets = an upload of the sequences "
                    "round r returned (round 5's way) instead of the consensus the library still holds in HBM")
    ap.add_argument("--no-quality", action="store_true", help="polish FASTA-like reads (no block qualities attached)")
    ap.add_argument("--replicas", action="store_true", help="N > 1: independent shard per GPU (weak scaling, no "
                    "collective) instead of one genome sharded across the ranks")
    ap.add_argument("--sharded", action="store_true", help="run the sharded code path even at N = 1 (one rank owning "
                    "every read and hash class): measures what the partition / regroup steps cost before any link traffic")
    args = ap.parse_args()
    args.errors = (0.04, 0.03, 0.03)
    if args.workload == "c4":
        args.genome = args.genome or 100_000_000
        args.read_len = args.read_len or 9000
        args.length_model = args.length_model or "lognormal"
    elif args.workload == "c5":
        args.genome = args.genome or 100_000_000
        args.read_len = args.read_len or 15000
        args.length_model = args.length_model or "normal"
        args.errors = (0.001, 0.002, 0.002)
        if args.coverage == 30.0:
            args.coverage = 40.0
        args.identity = 0.95 if args.identity is None else args.identity
    else:
        args.genome = args.genome or 5_000_000
        args.read_len = args.read_len or 10000
        args.length_model = args.length_model or "fixed"

    rank, world, local_rank = rdist.env_rank()

    import torch  # plumbing only: device selection, data generation, barrier, collectives
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dist = None
    # one rank per GPU over RCCL; a --sharded run launched by torch.distributed.run at world size 1 goes through the
    # process group as well (the nccl path on the one GPU of a test box)
    if world > 1 or (args.sharded and "RANK" in os.environ and "MASTER_PORT" in os.environ):
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

    sharded_mode = (world > 1 and not args.replicas) or args.sharded
    if world > 1 and args.replicas:  # independent shards: rank-dependent seeds would go here; same data is fine
        pass
    rs, drafts, t_gen = make_workload(args, local_rank)

    eng = hip.Engine(args.k, args.w, device=local_rank)
    peng = eng if (args.k, args.w) == (15, 5) else hip.Engine(15, 5, device=local_rank)  # racon maps with (15, 5)
    for kv in args.engine_option:
        name, value = kv.split("=")
        for x in {id(eng): eng, id(peng): peng}.values():
            x.set_option(name, int(value))
    t0 = time.time()
    reads = eng.upload(rs)  # H2D once; resident for every step
    preads = reads if peng is eng else peng.upload(rs)
    t_h2d = time.time() - t0
    if args.polish_rounds > 0 and not args.no_quality:
        # FASTQ reads at Phred 10 — biosoup's block_quality (mean of 64 bases, + 33) kept beside the packed
        # reads, so that racon's quality filter and quality weights are inside the timed rounds
        nblk = (rs.lengths.astype(np.int64) + 63) // 64
        qoff = np.zeros(rs.n + 1, dtype=np.uint64)
        np.cumsum(nblk, out=qoff[1:])
        preads.attach_quality((np.full(int(qoff[-1]), 33 + 10, dtype=np.uint8), qoff), block_shift=6)
    eng.set_timing(False)  # no per-stage host syncs inside the timed region
    peng.set_timing(False)
    eng.set_kernel_timing(not args.no_kernel_timing)
    peng.set_kernel_timing(not args.no_kernel_timing)

    comm = None
    if sharded_mode:
        from raven_amd import sharded
        comm = sharded.DeviceComm(dist, device="cuda", force=dist is not None)
    dev = torch.device("cuda", local_rank)
    own_reads, shard_laps = None, ({} if os.environ.get("RVN_SHARD_LAPS") else None)
    if sharded_mode:  # this rank's reads, resident for every step like `reads` of the single-GPU pass
        b = sharded.partition_reads(rs.lengths, world)
        own_reads = eng.upload(sharded.slice_reads(rs, int(b[rank]), int(b[rank + 1])))
    legs = {"overlap_s": 0.0, "polish_s": 0.0, "overlap_steps": [], "polish_steps": [], "poa_ms": 0.0, "poa_rounds": 0}
    last = {}

    def step(timed):
        t_a = time.perf_counter()
        if sharded_mode:
            res = sharded.find_overlaps_and_create_piles_sharded_dev(eng, rs, comm, dev, freq=args.freq, kmax=args.kmax,
                                                                     own=own_reads, laps=shard_laps if timed else None,
                                                                     fetch=False)  # results stay in HBM, as in the 1-GPU leg
            res["pass1"].close()
            last["overlap"] = {k: res[k] for k in ("lo", "hi", "occurrence", "stats")}
        else:
            p = eng.find_overlaps_and_create_piles(reads, freq=args.freq, kmax=args.kmax)  # synchronous
            t_x = time.perf_counter()
            if args.identity:
                # ConstructGraph's stages -5 .. -4 around the device calls (construct.cc:650-707): TrimAndAnnotatePiles on
                # the coverage in HBM, the identity filter of ResolveContainedReads on the kept lists, contained reads by
                # Raven's own overlap rules (host code of the reference: here the library's __host__ build of
                # overlap_rules.h through the product entry point rvn_overlap_update_and_type, timed apart as host_rules_s),
                # then the whole second pass
                t_s0 = time.perf_counter()
                ovl, off = p.overlaps()
                begin, end, median, invalid = p.trim_and_annotate(4)
                begin, end = (begin.astype(np.uint32) << 4), (end.astype(np.uint32) << 4)
                t_s1 = time.perf_counter()
                kept, koff = eng.filter_overlaps_by_identity(reads, ovl, off, begin, end, invalid, args.identity)
                t_s2 = time.perf_counter()
                upd, ok, ty = hip.overlap_update_and_type(kept, begin, end, invalid.astype(np.uint8))
                contained = np.zeros(rs.n, bool)
                contained[upd["lhs_id"][(ok == 1) & (ty == 1)]] = True
                contained[upd["rhs_id"][(ok == 1) & (ty == 2)]] = True
                inv2 = (invalid.astype(bool) | contained).astype(np.uint8)
                t_s3 = time.perf_counter()
                res2 = eng.find_overlaps_and_repetitive_regions(reads, begin, end, inv2, freq=args.freq, kmer_len=args.k,
                                                                identity=args.identity)
                t_s4 = time.perf_counter()
                if timed:
                    st5 = legs.setdefault("c5", {"fetch_trim_s": 0.0, "identity_filter_s": 0.0, "host_rules_s": 0.0,
                                                 "second_pass_s": 0.0})
                    st5["fetch_trim_s"] += t_s1 - t_s0
                    st5["identity_filter_s"] += t_s2 - t_s1
                    st5["host_rules_s"] += t_s3 - t_s2
                    st5["second_pass_s"] += t_s4 - t_s3
                    last["c5"] = {"pass1_overlaps": int(ovl.shape[0]), "kept_by_identity": int(kept.shape[0]),
                                  "invalid_or_contained": float(inv2.mean()), "pass2_overlaps": int(res2["overlaps"].shape[0])}
                del res2, ovl, kept, upd
            p.close()
            if os.environ.get("RVN_DEBUG_PASS1"):
                print("[bench] pass1 call %.1f ms, close %.1f ms" % ((t_x - t_a) * 1e3, (time.perf_counter() - t_x) * 1e3),
                      file=sys.stderr)
        t_b = time.perf_counter()
        cur = None
        n_windows = 0
        for rnd in range(args.polish_rounds):
            # round r + 1 polishes round r's output (raven::Polish, polish.cc:43-74): the library still has it in HBM
            if cur is None or sharded_mode or args.targets_through_host:
                targets = peng.upload_codes(drafts if cur is None else cur)
            else:
                try:
                    targets = peng.polish_output_as_reads([len(c) for c in cur])
                except (ValueError, hip.RavenHipError):  # (the engine gave its scratch back in between: the host's copy)
                    targets = peng.upload_codes(cur)
            if sharded_mode:
                cur, ratio = sharded.polish_round_sharded(peng, targets, preads, comm, targets.rs)
                st = {"n_windows": int(((targets.rs.lengths.astype(np.int64) + 499) // 500).sum())}
            else:
                cur, ratio, st = peng.polish_round(targets, preads)
                # windows the first attempt (32-column band, rows on lanes) handed on to the 64-column kernel / further
                st["poa_windows_to_64_columns"] = peng.poa_narrow_windows()
                st["poa_windows_to_128_columns"] = peng.poa_wide_windows()
            targets.close()
            n_windows += st["n_windows"]
            last["polish"] = st
            if timed and "poa_ms" in st:
                legs["poa_ms"] += float(st["poa_ms"])
                legs["poa_rounds"] += 1
            last["ratio"] = float(np.mean(ratio)) if len(ratio) else 0.0
        t_c = time.perf_counter()
        if timed:
            legs["overlap_s"] += t_b - t_a
            legs["polish_s"] += t_c - t_b
            legs["overlap_steps"].append(round(t_b - t_a, 4))
            legs["polish_steps"].append(round(t_c - t_b, 4))
            last["n_windows"] = n_windows

    for _ in range(args.warmup):
        step(False)
    eng.reset_stats()
    if peng is not eng:
        peng.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0

    units = float(rs.total_bases) if (world == 1 or not sharded_mode) else float(rs.total_bases) / world
    dt, total_bases = rdist.aggregate(dt, units, dist, device="cuda")
    if sharded_mode:
        total_bases = float(rs.total_bases)

    counters_raw = eng.counters()
    kms = {}
    if not args.no_kernel_timing:
        kms = dict(eng.kernel_ms())
        if peng is not eng:
            for k2, v in peng.kernel_ms().items():
                a = kms.get(k2, (0.0, 0))
                kms[k2] = (a[0] + v[0], a[1] + v[1])
    poa_cells = peng.poa_cells()

    # through-the-boundary time of the overlap pass, once, outside the timed region: what a caller that hands over host
    # buffers and wants host vectors back pays (upload + pass + fetch of piles and overlap lists); the polishing rounds
    # above already include their uploads (draft contigs) and fetches (consensus)
    boundary = None
    if rank == 0 and not sharded_mode:
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        r2 = eng.upload(rs)
        tb1 = time.perf_counter()
        p2 = eng.find_overlaps_and_create_piles(r2, freq=args.freq, kmax=args.kmax)
        tb2 = time.perf_counter()
        pd, _ = p2.piles()
        po, _ = p2.overlaps()
        tb3 = time.perf_counter()
        boundary = {"upload_s": round(tb1 - tb0, 4), "pass_s": round(tb2 - tb1, 4), "fetch_s": round(tb3 - tb2, 4),
                    "fetched_bytes": int(pd.nbytes + po.nbytes),
                    "overlap_gbase_per_s": round(rs.total_bases / (tb3 - tb0) / 1e9, 3)}
        # the annotation the reference runs right after the pass (TrimAndAnnotatePiles, construct.cc:123-152), on the
        # coverage still in HBM: FindValidRegion(4) + FindMedian, then FindChimericRegions of the valid piles
        tb4 = time.perf_counter()
        _, _, _, inv = p2.trim_and_annotate(4)
        tb5 = time.perf_counter()
        p2.find_chimeric_regions(inv)
        tb6 = time.perf_counter()
        boundary["trim_and_median_s"] = round(tb5 - tb4, 4)
        boundary["find_chimeric_regions_s"] = round(tb6 - tb5, 4)
        p2.close()
        del pd, po
        if r2 is not reads:
            r2.close()
    # input path (rvn_reads_load: gz FASTQ -> inflate pool -> page-locked text slabs -> record scan -> text cut and packed
    # on the device), on a FASTQ of the first reads of this workload (Phred-10 qualities), outside the timed region and
    # bounded so that the default run stays within minutes: once as blocked gzip (bgzip's multi-member form: the pool
    # inflates members in parallel) and once as ONE gzip member (one deflate stream = one thread, front to back)
    load_stats = None
    if rank == 0 and not sharded_mode and args.load_bases > 0:
        import gzip
        import tempfile
        from raven_amd import seqio as _seqio
        tl0 = time.perf_counter()
        n_take, acc = 0, 0
        while n_take < rs.n and acc < args.load_bases:
            acc += int(rs.lengths[n_take])
            n_take += 1
        recs = []
        for i in range(n_take):
            sq = rs.inflate(i)
            recs.append(b"@r%d\n" % i + sq + b"\n+\n" + b"+" * len(sq) + b"\n")
        text = b"".join(recs)
        del recs
        tmpd = tempfile.mkdtemp(prefix="rvn_load_")
        variants = {}
        for tag, blob in (("bgzf", _seqio.bgzf_compress(text, 1)), ("single_member", gzip.compress(text, 1))):
            path = os.path.join(tmpd, "reads_%s.fastq.gz" % tag)
            with open(path, "wb") as f:
                f.write(blob)
            tl1 = time.perf_counter()
            lr = eng.load(path)
            t_load = time.perf_counter() - tl1
            st = lr.load_stats
            variants[tag] = {"bases": int(st["n_bases"]), "reads": int(st["n_sequences"]), "gz_bytes": len(blob),
                             "load_s": round(t_load, 3), "scan_s": round(st["parse_s"], 3), "device_s": round(st["device_s"], 3),
                             "inflate_threads": st["inflate_threads"], "members": st["members"], "streaming": st["streaming"],
                             "load_gbase_per_s": round(st["n_bases"] / t_load / 1e9, 4)}
            lr.close()
            os.remove(path)
        os.rmdir(tmpd)
        del text
        load_stats = dict(variants["bgzf"])
        load_stats["single_member"] = variants["single_member"]
        load_stats["files_written_in_s"] = round(time.perf_counter() - tl0 - variants["bgzf"]["load_s"]
                                                 - variants["single_member"]["load_s"], 1)
        load_stats["note"] = ("blocked gzip (bgzip): members inflated by a pool of host threads into page-locked slabs, one "
                              "memchr pass finds the records, the device cuts and packs; single_member: one deflate "
                              "stream, decoded front to back by one thread (this library's inflate_fast.h, three helper "
                              "threads checksum and place its output; zlib only if that attempt raises a doubt)")

    if rank == 0 and shard_laps:
        print("[bench] sharded pass laps (s, summed over the timed steps):", {k: round(v, 4) for k, v in shard_laps.items()},
              file=sys.stderr)
    if rank == 0:
        steps = max(args.steps, 1)
        counters = {k: v // steps for k, v in counters_raw.items()}
        val_bytes = 4 if 2 * eng.k < 32 else 8
        kernels, roofline, roofline_hbm, roofline_poa, roofline_nw = {}, None, None, None, None
        roofline_hbm_all = []  # the four HBM-bound kernels with the largest shares (match_count among them)
        if kms:
            tot = sum(v[0] for v in kms.values())
            for name, (ms, la) in sorted(kms.items(), key=lambda x: -x[1][0]):
                if la:
                    kernels[name] = {"ms_per_step": round(ms / steps, 4), "launches_per_step": la / steps,
                                     "avg_launch_ms": round(ms / la, 5), "share": round(ms / tot, 4) if tot else None}
                    if name == "nw_traceback":
                        # launched on streams of their own beside the sweeps of the next chunk: these launch times
                        # overlap other sites', the stage's wall time is last_polish_round.align_ms
                        kernels[name]["overlaps_other_sites"] = True
            # dominant kernel of the WHOLE step: the banded POA kernel (integer VALU bound)
            dom = next(iter(kernels), None)
            site = "poa_rows" if kms.get("poa_rows", (0, 0))[1] else "poa_banded"
            if site in kms and kms[site][1] and poa_cells["cells_full"] and legs["poa_rounds"]:
                # The window-consensus stage of a polishing round = ONE launch of poa4.hip's persistent kernel (site
                # "poa_rows") + the poa2.hip launch(es) for the windows it hands on (site "poa_banded").  The roofline is the
                # dominant KERNEL's: algorithmic cells of a round / the average duration of the persistent kernel's launch
                # (HIP events around that launch on the engine's stream: the figure rocprofv3's kernel statistics must agree
                # with); the whole stage (events around the batch, poa2 included) is priced beside it.
                ms_k, la_k = kms[site]
                cells = poa_cells["cells_full"] / max(poa_cells["calls"], 1) * 1.0  # per polishing round
                kern_s = ms_k / la_k / 1e3 if site == "poa_rows" else legs["poa_ms"] / legs["poa_rounds"] / 1e3
                stage_s = legs["poa_ms"] / legs["poa_rounds"] / 1e3
                cells_per_launch = cells
                achieved_tops = cells_per_launch * POA_MIN_OPS_PER_CELL / kern_s / 1e12
                banded = poa_cells["cells_banded"] / max(poa_cells["calls"], 1)
                pmc_site = "poa4_persistent" if site == "poa_rows" else "poa_banded"
                roofline_poa = {"bound": "valu", "kernel": "poa4_persistent_kernel" if site == "poa_rows" else "poa_banded",
                            "achieved": round(achieved_tops, 3), "peak": round(VALU_PEAK_LANE_OPS / 1e12, 1),
                            "unit": "T lane-ops/s", "frac": round(achieved_tops * 1e12 / VALU_PEAK_LANE_OPS, 4),
                            "traffic": pmc_traffic(pmc_site), "traffic_raw": pmc_traffic_raw(pmc_site),
                            "algorithmic_cells_per_launch": int(cells_per_launch),
                            "algorithmic_ops_per_cell": POA_MIN_OPS_PER_CELL,
                            "gcups_algorithmic": round(cells_per_launch / kern_s / 1e9, 1),
                            "gcups_banded_computed": round(banded / kern_s / 1e9, 1),
                            "frac_on_computed_cells": round(banded * POA_MIN_OPS_PER_CELL / kern_s / VALU_PEAK_LANE_OPS, 4),
                            "avg_launch_ms": round(kern_s * 1e3, 3),
                            "launches_per_round": la_k / max(poa_cells["calls"], 1),
                            "stage": {"ms_per_round": round(stage_s * 1e3, 3),
                                      "frac": round(cells * POA_MIN_OPS_PER_CELL / stage_s / VALU_PEAK_LANE_OPS, 4),
                                      "traffic": pmc_traffic("poa_banded"), "traffic_raw": pmc_traffic_raw("poa_banded"),
                                      "handed_on_kernel_launches_per_round": kms.get("poa_banded", (0, 0))[1] / max(poa_cells["calls"], 1),
                                      "share_of_step": round(legs["poa_ms"] / (dt * 1e3), 3)},
                            "note": "algorithmic cells = graph rows x layer length of every layer alignment (what "
                                    "spoa's full NW computes); the first attempt computes a 32-column band of them "
                                    "(poa4.hip: one persistent kernel per polishing round = the launch priced here), what it "
                                    "hands on a 64-column band (poa2.hip: in 'stage').  avg_launch_ms = HIP events around the "
                                    "kernel's launch; rocprofv3's average for poa4_persistent_kernel is in "
                                    "profiles/r06_kernel_stats.csv."}
            if dom in ("poa_banded", "poa_rows") or legs["poa_ms"] > 0.4 * dt * 1e3:
                roofline = roofline_poa
            if "nw_forward" in kms and kms["nw_forward"][1] and last.get("polish", {}).get("align_band_cells"):
                # the alignment-path sweep (Myers bit-vector band, racon's edlib NW): integer VALU bound as well.  A round
                # launches it once per kernel variant and chunk (different sizes): cells of the round / summed launch time
                ms, la = kms["nw_forward"]
                rounds_timed = max(args.steps * args.polish_rounds, 1)
                cells_per_round = last["polish"]["align_band_cells"]
                s_summed = ms / rounds_timed / 1e3
                # (since round 6 sweep launches of few waves run beside the main stream's: summed launch times count that
                # time twice; the stage's wall clock — pilot, sweeps, the last walk's tail — is the basis that cannot flatter)
                s_per_round = float(last["polish"].get("align_ms", 0.0)) / 1e3 or s_summed
                achieved_tops = cells_per_round * NW_MIN_OPS_PER_CELL / s_per_round / 1e12
                roofline_nw = {"bound": "valu", "kernel": "nw_forward", "achieved": round(achieved_tops, 3),
                            "peak": round(VALU_PEAK_LANE_OPS / 1e12, 1), "unit": "T lane-ops/s",
                            "frac": round(achieved_tops * 1e12 / VALU_PEAK_LANE_OPS, 4), "traffic": pmc_traffic("nw_forward"),
                            "algorithmic_cells_per_round": int(cells_per_round),
                            "launches_per_round": la / rounds_timed, "ms_per_round": round(s_per_round * 1e3, 3),
                            "algorithmic_ops_per_cell": round(NW_MIN_OPS_PER_CELL, 3),
                            "gcups_band": round(cells_per_round / s_per_round / 1e9, 1),
                            "summed_launch_ms_per_round": round(s_summed * 1e3, 3),
                            "frac_on_summed_launch_time": round(cells_per_round * NW_MIN_OPS_PER_CELL / s_summed / VALU_PEAK_LANE_OPS, 4),
                            "traceback_ms_per_round": round(kms.get("nw_traceback", (0.0, 0))[0] / rounds_timed, 3),
                            "kernel_ms_share": round(ms / tot, 3) if tot else None,
                            "note": "algorithmic cells = cells of the Ukkonen band of every sweep the round ran (one per "
                                    "alignment; the few repeats with a doubled threshold and the 1024-job pilot "
                                    "included), each swept ONCE; achieved = cells x 50/64 lane-ops / the alignment stage's "
                                    "wall time of the last round (align_ms: pilot + sweeps + the tail of the last walk); "
                                    "summed_launch_ms_per_round = HIP events around every sweep launch on its stream, which "
                                    "overlap since launches of few waves run beside the main stream's.  The walk "
                                    "(nw_traceback) runs on other streams beside the sweeps; its launch times overlap them."}
                kernels["nw_forward"]["overlaps_other_sites"] = True
            if dom == "nw_forward":
                roofline = roofline_nw
            # the dominant HBM-bound kernel (second entry).  Kernels whose launches PARTITION a step's work (one launch per
            # flush window / interval class) are priced step against step: algorithmic bytes of the step / their summed
            # launch time; the others (one launch = one pass over everything, e.g. a radix digit pass) per launch
            partitioned = {"match_count", "match_emit", "seg_sort_group", "seg_sort_pos", "chain"}
            for name in kernels:
                b = algorithmic_bytes(name, counters, val_bytes)
                if b:
                    launches = kms[name][1] / steps
                    if name in partitioned:
                        t_s = kms[name][0] / steps / 1e3
                        basis = "step"
                    else:
                        t_s = kms[name][0] / kms[name][1] / 1e3
                        basis = "launch"
                    achieved = b / t_s / 1e9
                    # traffic on the SAME basis as algorithmic_bytes: the PMC file holds bytes per launch
                    per_launch, per_launch_raw = pmc_traffic(name), pmc_traffic_raw(name)
                    scale = launches if basis == "step" else 1.0
                    entry = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                                    "traffic": int(per_launch * scale) if per_launch else None,
                                    "traffic_raw": int(per_launch_raw * scale) if per_launch_raw else None,
                                    "algorithmic_bytes": int(b), "per": basis, "seconds": round(t_s, 6),
                                    "traffic_over_algorithmic": round(per_launch * scale / b, 2) if per_launch else None,
                                    "launches_per_step": launches,
                                    "kernel_ms_share": round(kms[name][0] / tot, 3) if tot else None,
                                    "note": "traffic, traffic_raw and algorithmic_bytes are all per %s (the PMC file's bytes "
                                            "per launch x %g launches)" % (basis, scale)}
                    if entry["traffic_raw"]:
                        entry["traffic_raw_over_algorithmic"] = round(entry["traffic_raw"] / b, 2)
                    if roofline_hbm is None:
                        roofline_hbm = entry  # the HBM-bound kernel with the largest share of the step
                    roofline_hbm_all.append(entry)
                    if len(roofline_hbm_all) >= 4:
                        break
            if roofline is None:
                roofline = roofline_hbm
        ovl_s, pol_s = legs["overlap_s"] / steps, legs["polish_s"] / steps
        rounds = args.polish_rounds
        out = {
            "metric": "read Gbase/s through overlap+polish",
            "value": round(total_bases * args.steps / dt / 1e9, 4),
            # the same with what the boundary adds to every step when the caller hands over host buffers and takes the
            # pass's piles and overlap lists back as host vectors (upload + fetch, measured once outside the timed region)
            "value_through_boundary": (round(total_bases / (dt / steps + boundary["upload_s"] + boundary["fetch_s"]) / 1e9, 4)
                                       if boundary else None),
            "unit": "Gbase/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3),
            "higher_is_better": True,
            # N > 1 shards ONE genome over the ranks (total work fixed): the N = 1 line is the first point of that curve;
            # --replicas (independent shard per GPU) is the weak-scaling variant
            "scaling": "weak" if (world > 1 and args.replicas) else "strong",
            "vs_baseline": None,
            "dtype": "u32" if val_bytes == 4 else "u64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[%d]%s: synthetic %.0f Mb genome, %gx reads (%s %d bp; %.1f%% sub, %.1f%% ins, "
                            "%.1f%% del), k=%d w=%d, FindOverlapsAndCreatePiles%s + -p %d (racon rounds on %.0f Mb draft contigs "
                            "with 2.6%% errors; %s)" % (
                                3 if args.workload == "c4" else (4 if args.workload == "c5" else (2 if rounds else 1)),
                                " on one GPU" if world == 1 and args.workload in ("c4", "c5") else "",
                                args.genome / 1e6, args.coverage, args.length_model, args.read_len,
                                100 * args.errors[0], 100 * args.errors[1], 100 * args.errors[2], args.k, args.w,
                                (" + TrimAndAnnotatePiles + identity filter %.2f + FindOverlapsAndRepetetiveRegions" % args.identity)
                                if args.identity else "", rounds, args.contig / 1e6,
                                "reads WITHOUT qualities: the FASTA variant" if args.no_quality else
                                "reads carry biosoup block qualities at Phred 10: the FASTQ variant"),
                "reads": rs.n, "read_bases": rs.total_bases, "freq": args.freq, "kmax": args.kmax,
                "draft_contigs": len(drafts),
                "parallelism": ("one genome sharded over %d GPUs: reads by pile, minimizers by hash class, 3 all-to-all + "
                                "1 all-reduce per flush window over RCCL; polishing windows by range" % world)
                if sharded_mode else ("independent replica per GPU (no data-path collective)" if world > 1 else "1 GPU"),
                "collectives": ("nccl" if dist is not None else None),
            },
            "exchange_bytes_per_step": (int(comm.bytes_sent // max(args.steps + args.warmup, 1)) if comm is not None else None),
            "legs": {"overlap_s_per_step": round(ovl_s, 4), "polish_s_per_step": round(pol_s, 4),
                     "overlap_s_of_each_step": legs["overlap_steps"], "polish_s_of_each_step": legs["polish_steps"],
                     "overlap_gbase_per_s": round(rs.total_bases / ovl_s / 1e9, 3) if ovl_s else None,
                     "polish_gbase_per_s_per_round": round(rs.total_bases / (pol_s / rounds) / 1e9, 3) if rounds and pol_s else None,
                     "windows_per_s": round(last.get("n_windows", 0) / pol_s, 1) if pol_s else None,
                     "polished_ratio": last.get("ratio")},
            "c5_stages": ({"seconds_per_step": {k: round(v / steps, 4) for k, v in legs["c5"].items()}, **last.get("c5", {}),
                           "identity": args.identity} if "c5" in legs else None),
            "overlaps_per_s": round(counters["overlaps"] / ovl_s, 1) if ovl_s else None,
            "counters_per_step": counters,
            "last_polish_round": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in last.get("polish", {}).items()},
            "roofline": roofline,
            "roofline_traffic_source": traffic_provenance(),
            "roofline_hbm": roofline_hbm,
            "roofline_hbm_kernels": roofline_hbm_all,
            "roofline_poa": roofline_poa if roofline is not roofline_poa else None,
            "roofline_nw": roofline_nw if roofline is not roofline_nw else None,
            "kernels": dict(list(kernels.items())[:16]),
            "host": {"gen_s": round(t_gen, 2), "h2d_s": round(t_h2d, 3),
                     "h2d_inclusive_gbase_s": round(rs.total_bases / (dt / steps + t_h2d) / 1e9, 4),
                     "overlap_pass_through_boundary": boundary, "input_path": load_stats},
            "cpu_baseline": None,
        }
        if sharded_mode:
            out["rank0"] = last.get("overlap")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, os.cpu_count() or 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
