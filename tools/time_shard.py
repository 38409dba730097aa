#!/usr/bin/env python3
"""Wall time of the individual calls of the sharded pass's first stage at one rank (where does the host time go)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raven_amd import hip, sharded, synth  # noqa: E402

dev = torch.device("cuda", 0)
g = synth.make_genome_torch(5_000_000, seed=11, device=dev)
rs, _ = synth.make_reads_torch(g, 30, 10000, length_model="fixed", sub=0.04, ins=0.03, dele=0.03, seed=12)
eng = hip.Engine(15, 5)
own = eng.upload(rs)
eng.set_timing(False)
eng.set_kernel_timing(bool(os.environ.get("KT")))
comm = sharded.DeviceComm(None, device="cuda")
i64 = dict(dtype=torch.int64, device=dev)


def T(name, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print("%-28s %8.3f ms" % (name, (time.perf_counter() - t) * 1e3))
    return r


for it in range(3):
    print("-- iteration", it)
    n = T("shard_sketch_count", lambda: eng.shard_sketch_count(own, index_minhash=False))
    val, org, val_p, org_p = T("4 x torch.empty", lambda: [torch.empty(n, **i64) for _ in range(4)])
    T("sketch_fetch_dev", lambda: eng.shard_sketch_fetch_dev(val.data_ptr(), org.data_ptr()))
    cnt = T("split_minimizers", lambda: eng.shard_split_minimizers_dev(val.data_ptr(), org.data_ptr(), n, 1, val_p.data_ptr(), org_p.data_ptr()))
    nf = T("count_flagged", lambda: eng.shard_count_flagged_dev(org_p.data_ptr(), n))
    T("index_build_dev", lambda: eng.shard_index_build_dev(val_p.data_ptr(), org_p.data_ptr(), n, False, nf))
    T("key_histogram", lambda: eng.shard_key_histogram())
    del val, org, val_p, org_p
    T("fused pass for comparison", lambda: eng.find_overlaps_and_create_piles(own).close())
    laps = {}
    T("whole sharded pass", lambda: sharded.find_overlaps_and_create_piles_sharded_dev(eng, rs, comm, dev, own=own))
    T("whole sharded pass (laps)", lambda: sharded.find_overlaps_and_create_piles_sharded_dev(eng, rs, comm, dev, own=own, laps=laps))
    print({k: round(v * 1e3, 2) for k, v in laps.items()})
