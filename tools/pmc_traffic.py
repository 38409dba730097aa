#!/usr/bin/env python3
"""Combine the FETCH_SIZE and WRITE_SIZE rocprofv3 --pmc passes into per-kernel HBM bytes per launch:
    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> profiles/r05_pmc_traffic.json [rounds] [calibration.json]
Per kernel site BOTH readings are kept:
    hbm_bytes_raw_per_launch = FETCH_SIZE * 1024 + WRITE_SIZE * 1024          the counters as rocprofv3 reports them
    hbm_bytes_per_launch     = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024      the guide's gfx950 correction (MI355X_MICROARCH.md
                               HBM section: FETCH_SIZE reports half the bytes of WIDE COALESCED reads) — an upper reading for
                               kernels that gather: tools/pmc_calibrate.sh measures what the counters report for streams,
                               4-byte / 16-byte gathers and scattered 2-byte stores on this box, and its factors are copied
                               into the file ("calibration").
Template instances of one kernel site (bench.py's kernel names) are pooled: bytes per launch = sum over the instances /
number of launches."""
import collections
import csv
import json
import re
import sys

SITES = [  # (regex on the demangled kernel name, bench.py kernel-site name)
    (r"nw_sweep_kernel", "nw_forward"), (r"nw_trace_kernel", "nw_traceback"), (r"poa2_kernel", "poa2"),
    (r"poa4_persistent_kernel", "poa4_persistent"),
    (r"\bpoa_kernel", "poa"), (r"chain_small_kernel", "chain_small"), (r"chain_kernel", "chain"),
    (r"rs_downsweep_kernel", "rs_downsweep"), (r"rs_upsweep_kernel", "rs_upsweep"),
    (r"sketch_kernel<[^>]*false>", "sketch_count"), (r"sketch_kernel<[^>]*true>", "sketch_write"),
    (r"join_kernel<false>", "join_count"), (r"join_kernel<true>", "join_emit"), (r"seg_sort_group_(lds|big)_kernel", "seg_sort_group"),
    (r"seg_sort_pos_(lds|big)_kernel", "seg_sort_pos"),
    (r"minhash_select_kernel", "minhash_select"), (r"unique_kernel", "unique"), (r"heads_kernel", "heads"),
    (r"match_count_kernel", "match_count"), (r"match_emit_kernel", "match_emit"), (r"table_kernel", "table"),
    (r"add_layers_kernel", "add_layers"), (r"ed_banded_kernel", "edit_distance"), (r"ed_lane_kernel", "edit_distance_lane"),
]


def site_of(name):
    name = name.replace("(anonymous namespace)::", "")
    for rx, site in SITES:
        if re.search(rx, name):
            return site
    m = re.match(r"(?:void )?(?:rvn::)?([A-Za-z_0-9]+)", name)
    return m.group(1) if m else name


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = agg[site_of(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def kernel_source_hash():
    """the same hash bench.py computes: what this profile is valid for"""
    import glob
    import hashlib
    import os
    h = hashlib.sha1()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raven_amd", "csrc")
    for fn in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, "rb").read())
    return h.hexdigest()


def main(fetch_csv, write_csv, out, rounds=None, calibration=None):
    f, w = load(fetch_csv), load(write_csv)
    kernels = {}
    for k in sorted(set(f) | set(w)):
        nf, fe = f.get(k, [0, 0.0])
        nw, wr = w.get(k, [0, 0.0])
        fe_l = fe * 1024 / max(nf, 1)
        wr_l = wr * 1024 / max(nw, 1)
        kernels[k] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_size_bytes_per_launch": int(fe_l),
                      "write_size_bytes_per_launch": int(wr_l), "hbm_bytes_raw_per_launch": int(fe_l + wr_l),
                      "hbm_bytes_per_launch": int(2 * fe_l + wr_l)}
    # the window-consensus stage of a polishing round = one launch of poa4.hip's persistent kernel + poa2.hip for what the
    # 32-column band hands on: its traffic is the sum over both / the polishing rounds of the profiled command
    stage = {"fetch": 0.0, "write": 0.0}
    for k in set(f) | set(w):
        if k.startswith("poa4_") or k == "poa2":
            stage["fetch"] += f.get(k, [0, 0.0])[1] * 1024
            stage["write"] += w.get(k, [0, 0.0])[1] * 1024
    if rounds:
        r = float(rounds)
        kernels["poa_banded"] = {"launches_fetch_pass": int(r), "launches_write_pass": int(r),
                                 "fetch_size_bytes_per_launch": int(stage["fetch"] / r),
                                 "write_size_bytes_per_launch": int(stage["write"] / r),
                                 "hbm_bytes_raw_per_launch": int((stage["fetch"] + stage["write"]) / r),
                                 "hbm_bytes_per_launch": int((2 * stage["fetch"] + stage["write"]) / r),
                                 "note": "one 'launch' = the launches of one polishing round (poa4_persistent + poa2 escalations)"}
    cal = None
    if calibration:
        try:
            cal = {k: {x: y for x, y in v.items() if not x.endswith("_KiB")} for k, v in json.load(open(calibration))["patterns"].items()}
        except (OSError, ValueError, KeyError):
            cal = None
    json.dump({"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py "
                         "--no-cpu-baseline`; raw = FETCH_SIZE*1024 + WRITE_SIZE*1024, corrected = 2*FETCH_SIZE*1024 + "
                         "WRITE_SIZE*1024 per launch (gfx950: FETCH_SIZE counts 128-B requests of wide coalesced reads as 64 B, "
                         "MI355X_MICROARCH.md HBM section); 'calibration' = counter bytes / known bytes of tools/pmc_calibrate.hip's "
                         "access patterns on the same box",
               "kernel_source_sha1": kernel_source_hash(), "calibration": cal, "kernels": kernels},
              open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main(*sys.argv[1:4], rounds=(sys.argv[4] if len(sys.argv) > 4 else None), calibration=(sys.argv[5] if len(sys.argv) > 5 else None))
