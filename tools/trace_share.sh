#!/bin/bash
# Kernel timeline of the alignment stage of ONE rank's share of a C4 polishing round at 8 virtual ranks (the last share
# tools/rank_share.py runs) -> gpurun_out/<tag>_share_nw_timeline.csv.   usage: trace_share.sh <tag> [rank_share.py arguments]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06g}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_share_tl -o t -- python $R/tools/rank_share.py --ranks 8 --rounds 1 --polish-only "$@" > $R/gpurun_out/${TAG}_share_trace.json 2> $R/gpurun_out/${TAG}_share_tl.err
F=$(find $R/gpurun_out/${TAG}_share_tl -name '*kernel_trace.csv' | head -1)
python - "$F" $R/gpurun_out/${TAG}_share_nw_timeline.csv <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(r"nw_\w+_kernel", r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = 0
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - max(int(r["End_Timestamp"]) for r in rows[last:i]) > 50e6:
        last = i
rows = rows[last:]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
with open(sys.argv[2], "w") as f:
    f.write("kernel,queue,start_ms,end_ms,dur_ms,grid\n")
    for r in rows:
        n = re.search(r"nw_\w+_kernel(<[^>]*>)?", r["Kernel_Name"]).group(0).replace(",", ";")
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        f.write("%s,%s,%.2f,%.2f,%.2f,%s\n" % (n, r.get("Queue_Id", ""), s / 1e6, e / 1e6, (e - s) / 1e6, r.get("Grid_Size", r.get("Grid_Size_X", ""))))
PY
rm -rf $R/gpurun_out/${TAG}_share_tl
cat $R/gpurun_out/${TAG}_share_nw_timeline.csv
