#!/bin/bash
# correctness vs poa2 + throughput + per-kernel times of poa4 (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-24576}
cd $R
timeout 100 python tools/check_poa4.py 400 9 2>&1 | tail -2
RVN_POA_STATS=1 RVN_POA_MODES=${2:-9} timeout 250 python tools/bench_poa.py $N 20 > gpurun_out/r04_bx.json 2> gpurun_out/r04_bx.err
python - <<PY
import json
for l in open("gpurun_out/r04_bx.json"):
    d=json.loads(l); print(d["run"], round(d["device_ms"],1), round(d["windows_per_s"]), {k:round(v/1e9,1) for k,v in d["phase_cycles"].items()}, d["status_counts"], d.get("identical_to_cpu"))
PY
cd /tmp && export TMPDIR=/tmp && RVN_POA_MODE=9 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_kt -o p -- python $R/tools/bench_poa.py $N 0 > /dev/null 2>&1
F=$(find $R/gpurun_out/r04_kt -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in list(csv.DictReader(open("$F")))[:8]:
    import re as _re; _m=_re.search(r"poa[234]_\w+", r["Name"]); n=_m.group(0) if _m else r["Name"][:30]
    print(n, r["Calls"], "total_ms", round(int(r["TotalDurationNs"])/1e6,2), "max_ms", round(int(r["MaxNs"])/1e6,3))
PY
cp $F $R/gpurun_out/r04_kt_stats.csv; rm -rf $R/gpurun_out/r04_kt
