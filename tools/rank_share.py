#!/usr/bin/env python3
"""What ONE rank of an N-GPU run of the metric's workload does, measured on the one GPU there is (no
scaling curve can be taken — every test box has one GPU — so the per-rank times and exchange volumes an 8-GPU run is made of
are measured, and the step time they add up to is predicted next to single-GPU / N).

    python tools/rank_share.py [--ranks 8] [--workload c4] [--genome N]  > profiles/r06_rank_share_c4.json

* Overlap pass: raven_amd.sharded.find_overlaps_and_create_piles_sharded_dev — the code `bench.py --gpus N` runs, unmodified —
  with N virtual ranks (threads, an engine each, all on this GPU) under a communicator that lets exactly ONE rank run at a
  time: a rank computes from one collective to the next alone on the GPU, its time is taken, the next rank runs.  The data
  path is the real one (partition kernels, regroup, merge); a collective here is a device copy between the ranks' tensors, so
  its TIME is not a link's: the bytes every rank sends are counted and priced at one xGMI link per peer (153 GB/s, the
  seven links of a rank in parallel, MI355X_MICROARCH.md) + 30 us per collective.
* Polishing rounds: raven_amd.sharded.polish_round_sharded's two splits with one engine — rank g's slice of the reads
  through rvn_polish_map_best, then rank g's window range through rvn_polish_round_range — one rank after the other, each
  timed; the all-gathers (best-overlap table, consensus pieces) are priced by their bytes.
* A bulk-synchronous step at N ranks = sum over the segments between collectives of the SLOWEST rank's time + the priced
  exchanges.  predicted efficiency = single-GPU step / (N x that).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (workload generator of the bench line)
from raven_amd import hip, sharded  # noqa: E402

LINK_GBS = 153.0       # one xGMI link, one direction (MI355X_MICROARCH.md)
COLLECTIVE_US = 30.0   # launch + rendezvous of one RCCL collective on a node


class Turns:
    """One rank at a time, in rank order; a rank that has returned is skipped."""

    def __init__(self, world):
        self.cv = threading.Condition()
        self.cur, self.world = 0, world
        self.done = [False] * world
        self.error = None

    def acquire(self, r):
        with self.cv:
            self.cv.wait_for(lambda: self.cur == r or self.error is not None)
            if self.error is not None:
                raise RuntimeError("another rank failed") from self.error

    def release(self, r, finished=False):
        with self.cv:
            self.done[r] = self.done[r] or finished
            if not all(self.done):
                nxt = (r + 1) % self.world
                while self.done[nxt]:
                    nxt = (nxt + 1) % self.world
                self.cur = nxt
            self.cv.notify_all()

    def fail(self, ex):
        with self.cv:
            self.error = ex
            self.cv.notify_all()


class TurnComm:
    """Interface of raven_amd.sharded.DeviceComm / tests.sharded_util.LocalComm; times the compute between collectives."""

    def __init__(self, turns, store, rank, world):
        self.t, self.store, self.rank, self.world = turns, store, rank, world
        self.bytes_sent = 0
        self.seq = 0
        self.segments = []     # seconds of compute before collective k (alone on the GPU)
        self.exchanges = []    # (kind, bytes sent to other ranks, seconds of the device copies standing in for the link)
        self._t0 = None

    def start(self):
        import torch
        self.t.acquire(self.rank)
        torch.cuda.synchronize()
        self._t0 = time.perf_counter()

    def finish(self):
        import torch
        torch.cuda.synchronize()
        self.segments.append(time.perf_counter() - self._t0)
        self.t.release(self.rank, finished=True)

    def _exchange(self, kind, parts, nbytes, clone):
        import torch
        torch.cuda.synchronize()
        self.segments.append(time.perf_counter() - self._t0)
        slot = self.store.setdefault(self.seq, [None] * self.world)
        slot[self.rank] = parts
        self.t.release(self.rank)
        self.t.acquire(self.rank)   # every other rank has run up to (and deposited for) this collective by now
        t1 = time.perf_counter()
        res = [clone(self.store[self.seq][s][self.rank]) for s in range(self.world)]
        torch.cuda.synchronize()
        sent = sum(nbytes(p) for i, p in enumerate(parts) if i != self.rank)
        self.exchanges.append((kind, int(sent), time.perf_counter() - t1))
        self.bytes_sent += sent
        self.seq += 1
        self._t0 = time.perf_counter()
        return res

    def all_to_all_v(self, parts, kind="all_to_all(host)"):
        return self._exchange(kind, parts, lambda p: p.nbytes, lambda p: np.array(p, copy=True))

    def all_to_all_t(self, parts, kind="all_to_all"):
        return self._exchange(kind, parts, lambda p: 8 * int(p.shape[0]), lambda p: p.clone())

    def all_to_all_flat_t(self, flat, send_lens):
        import torch
        send_lens = [int(x) for x in send_lens]
        parts = list(torch.split(flat[:sum(send_lens)], send_lens))
        res = self.all_to_all_t(parts)
        return (torch.cat(res) if res else flat[:0]), [int(x.shape[0]) for x in res]

    def all_reduce_sum(self, a):
        return np.sum(self.all_to_all_v([a] * self.world, kind="all_reduce"), axis=0)

    def all_gather_v(self, a):
        return np.concatenate(self.all_to_all_v([a] * self.world, kind="all_gather"))


def price(bytes_per_rank, world, n_collectives):
    """Seconds of an all-to-all in which the busiest rank sends `bytes_per_rank` to its world - 1 peers over one link each."""
    per_peer = bytes_per_rank / max(world - 1, 1)
    return per_peer / (LINK_GBS * 1e9) + n_collectives * COLLECTIVE_US * 1e-6


def overlap_pass_shares(rs, world, k, w, freq, kmax, reps=2):
    import torch
    dev = torch.device("cuda", 0)
    bounds = sharded.partition_reads(rs.lengths, world)
    engines = [hip.Engine(k, w) for _ in range(world)]
    owns = [engines[r].upload(sharded.slice_reads(rs, int(bounds[r]), int(bounds[r + 1]))) for r in range(world)]
    for e in engines:
        e.set_timing(False)
    out = None
    for rep in range(reps):   # (the first pass grows the engines' scratch: the last one is reported)
        turns, store = Turns(world), {}
        comms = [TurnComm(turns, store, r, world) for r in range(world)]
        stats = [None] * world

        def work(r):
            try:
                comms[r].start()
                res = sharded.find_overlaps_and_create_piles_sharded_dev(engines[r], rs, comms[r], dev, freq=freq, kmax=kmax,
                                                                         own=owns[r], fetch=False)
                res["pass1"].close()
                stats[r] = res["stats"]
                comms[r].finish()
            except BaseException as ex:  # noqa: BLE001
                turns.fail(ex)
                raise

        ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if turns.error is not None:
            raise turns.error
        n_seg = len(comms[0].segments)
        assert all(len(c.segments) == n_seg for c in comms)
        seg = np.array([c.segments for c in comms])                      # [rank, segment]
        exch_bytes = np.array([[x[1] for x in c.exchanges] for c in comms])  # [rank, collective]
        kinds = [x[0] for x in comms[0].exchanges]
        a2a = [i for i, kd in enumerate(kinds) if kd == "all_to_all"]
        exch_s = sum(price(float(exch_bytes[:, i].max()), world, 1) for i in range(len(kinds)))
        out = {"ranks": world, "collectives_per_pass": len(kinds), "collective_kinds": {kd: kinds.count(kd) for kd in set(kinds)},
               "compute_s_per_rank": [round(float(x), 4) for x in seg.sum(axis=1)],
               "critical_path_compute_s": round(float(seg.max(axis=0).sum()), 4),
               "bytes_sent_per_rank": [int(x) for x in exch_bytes.sum(axis=1)],
               "all_to_all_bytes_busiest_rank": int(exch_bytes[:, a2a].max(axis=0).sum()) if a2a else 0,
               "exchanges_priced_s": round(exch_s, 4),
               "device_copy_stand_in_s_per_rank": [round(sum(x[2] for x in c.exchanges), 4) for c in comms],
               "predicted_pass_s": round(float(seg.max(axis=0).sum()) + exch_s, 4),
               "stats_rank0": stats[0]}
    for o in owns:
        o.close()
    for e in engines:
        e.close()
    return out


def polish_shares(peng, preads, drafts, world, rounds):
    """Per round and rank: the read slice's mapping, then the window range's alignment + consensus."""
    out = []
    cur = None
    for rnd in range(rounds):
        targets = peng.upload_codes(drafts if cur is None else cur)
        lengths = targets.rs.lengths.astype(np.int64)
        n_win = int(((lengths + 499) // 500).sum())
        n_reads = preads.n
        map_s, tables = [], []
        for g in range(world):
            t0 = time.perf_counter()
            best, bt, _ = peng.polish_map_best(targets, preads, n_reads * g // world, n_reads * (g + 1) // world)
            map_s.append(time.perf_counter() - t0)
            tables.append((best, bt))
        best = np.concatenate([t[0] for t in tables])
        bt = np.concatenate([t[1] for t in tables])
        peng.polish_set_best(best, bt)
        range_s, pieces, windows, stage_ms = [], [], [], []
        for g in range(world):
            lo, hi = n_win * g // world, n_win * (g + 1) // world
            for rep in range(2):  # (the second call is the one a rank in its steady state makes: scratch at its size)
                peng.polish_set_best(best, bt)
                t0 = time.perf_counter()
                cons, nw, npol, st = peng.polish_round_range(targets, preads, lo, hi)
                dt = time.perf_counter() - t0
            range_s.append(dt)
            stage_ms.append({k: round(st[k], 1) for k in ("align_ms", "poa_ms", "host_ms", "total_ms")})
            pieces.append(cons)
            windows.append(hi - lo)
        cur = [np.concatenate([pieces[g][t] for g in range(world)]) for t in range(targets.n)]
        table_bytes = 40 * n_reads                       # all-gather of the best-overlap table: everybody gets everything
        cons_bytes = int(sum(len(c) for c in cur))
        gathers_s = price(table_bytes * (world - 1) / world, world, 1) + price(cons_bytes * (world - 1) / world, world, 4)
        out.append({"round": rnd + 1, "windows_per_rank": windows, "map_s_per_rank": [round(x, 4) for x in map_s],
                    "align_and_consensus_s_per_rank": [round(x, 4) for x in range_s], "stages_ms_per_rank": stage_ms,
                    "all_gather_bytes": {"best_overlap_table": table_bytes, "consensus": cons_bytes},
                    "gathers_priced_s": round(gathers_s, 5),
                    "predicted_round_s": round(max(map_s) + max(range_s) + gathers_s, 4)})
        targets.close()
    return out, cur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--workload", default="c4", choices=["c4", "c2"])
    ap.add_argument("--genome", type=int, default=None)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--engine-option", action="append", default=[], metavar="NAME=VALUE")
    ap.add_argument("--polish-only", action="store_true", help="the rounds' shares only (tools/trace_share.sh): no overlap pass, no prediction")
    a = ap.parse_args()
    global ENGINE_OPTIONS
    ENGINE_OPTIONS = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in a.engine_option]
    args = argparse.Namespace(genome=a.genome or (100_000_000 if a.workload == "c4" else 5_000_000), coverage=30.0,
                              read_len=10000, length_model="lognormal" if a.workload == "c4" else "fixed",
                              errors=(0.04, 0.03, 0.03), polish_rounds=a.rounds, contig=5_000_000)
    rs, drafts, t_gen = bench.make_workload(args, 0)
    world = a.ranks

    # ---- the single GPU's step on the same data (what bench.py times): pass + rounds, second of two steps ----
    eng = hip.Engine(15, 5)
    for name, value in ENGINE_OPTIONS:
        eng.set_option(name, value)
    reads = eng.upload(rs)
    nblk = (rs.lengths.astype(np.int64) + 63) // 64
    qoff = np.zeros(rs.n + 1, dtype=np.uint64)
    np.cumsum(nblk, out=qoff[1:])
    reads.attach_quality((np.full(int(qoff[-1]), 33 + 10, dtype=np.uint8), qoff), block_shift=6)
    eng.set_timing(False)
    single = {}
    for it in range(2):
        t0 = time.perf_counter()
        eng.find_overlaps_and_create_piles(reads, freq=0.001, kmax=32).close()
        t1 = time.perf_counter()
        cur, rounds_s = None, []
        for rnd in range(a.rounds):
            targets = eng.upload_codes(drafts if cur is None else cur)
            t2 = time.perf_counter()
            cur, _, st = eng.polish_round(targets, reads)
            rounds_s.append(time.perf_counter() - t2)
            targets.close()
        single = {"pass_s": round(t1 - t0, 4), "round_s": [round(x, 4) for x in rounds_s],
                  "step_s": round(time.perf_counter() - t0, 4), "windows_per_round": int(st["n_windows"])}
    ref_cons = cur

    # ---- the rounds' rank shares (same engine), then the pass with N engines ----
    eng.release_scratch()  # (a rank's scratch grows to a rank's sizes, not to the whole genome's)
    shares, cons = polish_shares(eng, reads, drafts, world, a.rounds)
    same = all(np.array_equal(x, y) for x, y in zip(cons, ref_cons))
    reads.close()
    eng.release_scratch()
    eng.close()
    import torch
    torch.cuda.empty_cache()
    if a.polish_only:
        print(json.dumps({"ranks": world, "polishing_rounds_at_n_ranks": shares, "sharded_rounds_equal_single_gpu_consensus": bool(same)}))
        return
    ov = overlap_pass_shares(rs, world, 15, 5, 0.001, 32)

    t_n = ov["predicted_pass_s"] + sum(r["predicted_round_s"] for r in shares)
    out = {"workload": "configs[3]: %d Mb genome, 30x, %d reads, %.2f Gbases; Phred-10 block qualities; %d polishing rounds"
                       % (args.genome // 1_000_000, rs.n, rs.total_bases / 1e9, a.rounds),
           "ranks": world, "single_gpu": single, "overlap_pass_at_n_ranks": ov, "polishing_rounds_at_n_ranks": shares,
           "sharded_rounds_equal_single_gpu_consensus": bool(same),
           "predicted_step_s_at_n_ranks": round(t_n, 4), "ideal_step_s": round(single["step_s"] / world, 4),
           "predicted_gbase_s_at_n_ranks": round(rs.total_bases / t_n / 1e9, 3),
           "predicted_strong_scaling_efficiency": round(single["step_s"] / (world * t_n), 3),
           "model": "bulk-synchronous: per segment between two collectives the slowest rank's measured time (alone on the GPU); "
                    "a collective = busiest rank's bytes / (world - 1) peers at %.0f GB/s per xGMI link + %.0f us" % (LINK_GBS, COLLECTIVE_US),
           "gen_s": round(t_gen, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
