#!/usr/bin/env python3
"""Per-kernel average of a rocprofv3 --pmc counter (csv output) -> csv.
    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv profiles/r01_pmc_fetch.csv"""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?(?:rvn::)?([A-Za-z_0-9]+(?:<[^(]*>)?)", name)
    return m.group(1) if m else name


def main(src, dst):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(src)):
        key = (short(r["Kernel_Name"]), r["Counter_Name"])
        a = agg.setdefault(key, [0, 0.0, 0, 0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[3] = max(a[3], int(r["Grid_Size"]))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "AvgCounterValue", "AvgCounterBytes(x1024)", "AvgDurationUs", "MaxGrid"])
        for (k, c), a in sorted(agg.items(), key=lambda x: -x[1][1]):
            w.writerow([k, c, a[0], "%.1f" % (a[1] / a[0]), "%.0f" % (a[1] / a[0] * 1024), "%.2f" % (a[2] / a[0] / 1e3), a[3]])
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
