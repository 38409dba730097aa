#!/bin/bash
# per-kernel times of tools/bench_poa.py under rocprofv3 --kernel-trace --stats (environment passes through)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-24576}
cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_kt -o p -- python $R/tools/bench_poa.py $N 0 > /dev/null 2>&1
F=$(find $R/gpurun_out/r04_kt -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv,re
for r in list(csv.DictReader(open("$F")))[:9]:
    m=re.search(r'poa[234]_\w+', r["Name"]); n=m.group(0) if m else r["Name"][:30]
    print(n, r["Calls"], "total_ms", round(int(r["TotalDurationNs"])/1e6,2), "avg_ms", round(float(r["AverageNs"])/1e6,3), "max_ms", round(int(r["MaxNs"])/1e6,3))
PY
cp $F $R/gpurun_out/${2:-r04_kt_stats}.csv; rm -rf $R/gpurun_out/r04_kt
