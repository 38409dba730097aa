#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts, per access pattern (tools/pmc_calibrate.hip).
# usage (on the GPU box): bash tools/pmc_calibrate.sh <out.json>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/pmc_calibration.json}
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calibrate $R/tools/pmc_calibrate.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$c -o p -- /tmp/pmc_calibrate > /dev/null 2> /tmp/cal_$c.err
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
table, probes = 8 << 30, 1 << 28
known = {"stream_read16": ("FETCH_SIZE", table), "gather4": ("FETCH_SIZE", probes * 4), "gather16": ("FETCH_SIZE", probes * 16),
         "stream_write16": ("WRITE_SIZE", table), "scatter2": ("WRITE_SIZE", probes * 2)}
out = {"table_bytes": table, "probes": probes, "unit": "counter value x 1024 B (rocprofv3 reports KiB)", "patterns": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/cal_%s/**/*counter_collection.csv" % c, recursive=True)
    if not f:
        continue
    for row in csv.DictReader(open(f[0])):
        if row.get("Counter_Name") != c:
            continue
        k = row["Kernel_Name"].split("(")[0]
        e = out["patterns"].setdefault(k, {})
        e.setdefault(c + "_KiB", []).append(float(row["Counter_Value"]))
for k, (c, b) in known.items():
    e = out["patterns"].get(k)
    if not e or c + "_KiB" not in e:
        continue
    v = e[c + "_KiB"]
    raw = sum(v) / len(v) * 1024.0
    e["known_bytes"] = b
    e["counter_bytes_raw"] = raw
    e["raw_over_known"] = round(raw / b, 4)
    if k.startswith("gather") or k.startswith("scatter"):
        n = probes
        e["raw_bytes_per_probe"] = round(raw / n, 2)
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: {x: y for x, y in v.items() if not x.endswith("_KiB")} for k, v in out["patterns"].items()}))
PY
