#!/bin/bash
# per-kernel times of bench.py (one timed step) under rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_ktb -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ${@} > $R/gpurun_out/r04_ktb_bench.json 2> $R/gpurun_out/r04_ktb_bench.err
F=$(find $R/gpurun_out/r04_ktb -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv,re
for r in list(csv.DictReader(open("$F")))[:16]:
    m=re.search(r'(poa[234]_\w+|nw_\w+|\w+_kernel)', r["Name"]); n=m.group(0) if m else r["Name"][:30]
    print(n, r["Calls"], "total_ms", round(int(r["TotalDurationNs"])/1e6,2), "avg_ms", round(float(r["AverageNs"])/1e6,3), "max_ms", round(int(r["MaxNs"])/1e6,3))
PY
cp $F $R/gpurun_out/r04_ktb_stats.csv; rm -rf $R/gpurun_out/r04_ktb
python - <<PY
import json
d=json.loads(open("$R/gpurun_out/r04_ktb_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ["value","ms_per_step"]}); print(d["legs"]); print(d["last_polish_round"])
PY
