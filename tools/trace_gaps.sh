#!/bin/bash
# Kernel timeline of one warm bench step: where the GPU is idle (no kernel of any stream running), per gap the kernels either side.
# usage: trace_gaps.sh <tag> [workload]   -> gpurun_out/<tag>_gaps.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}; W=${2:-c4}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_gaps_tl -o t -- python $R/bench.py --workload $W --no-cpu-baseline --no-kernel-timing --load-bases 0 --steps 1 --warmup 1 > $R/gpurun_out/${TAG}_gaps_bench.json 2> $R/gpurun_out/${TAG}_gaps.err
F=$(find $R/gpurun_out/${TAG}_gaps_tl -name '*kernel_trace.csv' | head -1)
python - "$F" > $R/gpurun_out/${TAG}_gaps.txt <<'PY'
import csv, sys, re
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", re.sub(r"rvn::|\(anonymous namespace\)::|void ", "", r["Kernel_Name"]))[:48]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last step = the second half of the run by time is not known: take the kernels after the last gap > 300 ms (data generation / warm-up boundary is host work)
t_end = max(e for _, e, _ in rows)
# find start of last step: last occurrence of the first overlap-pass kernel sequence: use the last 'sketch' launch group start preceded by >= 3 ms idle
busy_end = 0; gaps = []; prev = None
for s, e, n in rows:
    if prev is not None and s > busy_end:
        gaps.append((s - busy_end, busy_end, s, prevname, n))
    if e > busy_end:
        busy_end = e; prevname = n
    prev = 1
# restrict to the last 2.6 s of the trace (one warm step and a bit)
lo = t_end - int(float(__import__("os").environ.get("GAP_WINDOW_S", "2.45")) * 1e9)
sel = [g for g in gaps if g[1] >= lo]
tot = sum(g[0] for g in sel)
print("window: last 2.45 s of the trace; idle (no kernel running) %.1f ms in %d gaps" % (tot / 1e6, len(sel)))
for g in (sorted(sel, key=lambda g: g[1]) if __import__("os").environ.get("GAP_BY_TIME") else sorted(sel, reverse=True)[:40]):
    if g[0] < 1.0e6 and __import__("os").environ.get("GAP_BY_TIME"): continue
    print("%8.2f ms at %9.2f ms  after %-48s before %s" % (g[0] / 1e6, (g[1] - lo) / 1e6, g[3], g[4]))
PY
rm -rf $R/gpurun_out/${TAG}_gaps_tl
head -45 $R/gpurun_out/${TAG}_gaps.txt
