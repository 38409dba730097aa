#!/usr/bin/env python3
"""Combine the FETCH_SIZE and WRITE_SIZE rocprofv3 --pmc passes into per-kernel HBM bytes per launch:
    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> profiles/pmc_traffic.json
hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024  (gfx950 correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE
reports half the bytes of wide coalesced reads; WRITE_SIZE is taken as is, uncalibrated)."""
import collections
import csv
import json
import re
import sys

SITE = {  # kernel name -> bench.py kernel-site name
    "rs_downsweep_kernel<unsigned int, unsigned long>": "rs_downsweep", "rs_upsweep_kernel<unsigned int>": "rs_upsweep",
    "sketch_kernel<unsigned int, false>": "sketch_count", "sketch_kernel<unsigned int, true>": "sketch_write",
    "join_kernel<false>": "join_count", "join_kernel<true>": "join_emit", "seg_sort_off_kernel": "seg_sort_group",
    "seg_sort_be_kernel": "seg_sort_pos", "chain_kernel": "chain", "chain_small_kernel": "chain_small",
    "minhash_select_kernel<unsigned int>": "minhash_select", "unique_kernel<unsigned int>": "unique",
    "heads_kernel<unsigned int>": "heads", "match_count_kernel<unsigned int>": "match_count",
    "match_emit_kernel": "match_emit", "table_kernel<unsigned int>": "table",
}


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?(?:rvn::)?([A-Za-z_0-9]+(?:<[^(]*>)?)", name)
    return m.group(1) if m else name


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in agg.items()}


def main(fetch_csv, write_csv, out):
    f, w = load(fetch_csv), load(write_csv)
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fe, wr = f.get(k, 0.0) * 1024, w.get(k, 0.0) * 1024
        kernels[SITE.get(k, k)] = {"kernel": k, "fetch_size_bytes": int(fe), "write_size_bytes": int(wr),
                                   "hbm_bytes_per_launch": int(2 * fe + wr)}
    json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --steps 2`; "
                         "hbm = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH correction)", "kernels": kernels},
              open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main(*sys.argv[1:4])
