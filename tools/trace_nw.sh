#!/bin/bash
# kernel timeline of the alignment stage (start / end per dispatch, both streams) -> gpurun_out/<tag>_nw_timeline.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}; W=${2:-c4}
cd /tmp && export TMPDIR=/tmp
export RVN_POLISH_SKIP_POA=1
export RVN_LIB_PATH=$R/raven_amd/lib/libraven_hip_test.so  # (the switch above exists in the debug build only)
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_nw_tl -o t -- python $R/bench.py --workload $W --no-cpu-baseline --no-kernel-timing --load-bases 0 --steps 1 --warmup 1 > /dev/null 2> $R/gpurun_out/${TAG}_nw_tl.err
F=$(find $R/gpurun_out/${TAG}_nw_tl -name '*kernel_trace.csv' | head -1)
python - "$F" $R/gpurun_out/${TAG}_nw_timeline.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "nw_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST polishing round of the run (a warm one: the first round of a process also grows its buffers): rounds are
# separated by the other stages, i.e. by gaps of more than 100 ms between two alignment kernels
last = 0
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - max(int(r["End_Timestamp"]) for r in rows[last:i]) > 100e6:
        last = i
rows = rows[last:]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
with open(sys.argv[2], "w") as f:
    f.write("kernel,queue,start_ms,end_ms,dur_ms,grid\n")
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        import re
        n = re.search(r"nw_\w+_kernel(<[^>]*>)?", r["Kernel_Name"]).group(0).replace(",", ";")
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        f.write("%s,%s,%.2f,%.2f,%.2f,%s\n" % (n, r.get("Queue_Id", ""), s / 1e6, e / 1e6, (e - s) / 1e6, r.get("Grid_Size", r.get("Grid_Size_X", ""))))
PY
rm -rf $R/gpurun_out/${TAG}_nw_tl
cat $R/gpurun_out/${TAG}_nw_timeline.csv | head -90
