#!/bin/bash
# Same box, same binary: the alignment stage with every walk by one lane per alignment (engine option nw_group_walk = 1),
# by the default rule (the group of lanes for launches of few alignments) and by a group everywhere (2), at C4 and C2;
# + the GPU tests of the polishing round and the kernel timeline of the alignment stage.
# usage: walk_ab.sh <tag>   -> gpurun_out/<tag>_walk_*.json, <tag>_nw_timeline.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06g}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_polish.py -x -q -m gpu > gpurun_out/${TAG}_tests_polish.log 2>&1; tail -3 gpurun_out/${TAG}_tests_polish.log
line() {
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print(sys.argv[1], "no JSON line:", e); sys.exit(0)
lp = d.get("last_polish_round") or {}
k = d.get("kernels", {}).get("nw_traceback", {})
print(sys.argv[1].split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "polish_s", d["legs"]["polish_s_per_step"],
      "align_ms", lp.get("align_ms"), "poa_ms", lp.get("poa_ms"), "walk launches ms/step", k.get("ms_per_step"))
PY
}
for w in c2 c4; do
  for rep in 1 2; do
    for mode in 1 0 2; do
      [ $w = c4 ] && [ $rep = 2 ] && continue
      f=gpurun_out/${TAG}_walk_${w}_mode${mode}_${rep}.json
      timeout 600 python bench.py --workload $w --steps 2 --warmup 2 --no-cpu-baseline --load-bases 0 --engine-option nw_group_walk=$mode > $f 2> ${f%.json}.err
      line $f
    done
  done
done
bash tools/trace_nw.sh $TAG > /dev/null 2>&1
python - gpurun_out/${TAG}_nw_timeline.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tr = [r for r in rows if "trace" in r["kernel"]]
print("alignment stage: %.1f ms from the first to the last kernel; walks:" % max(float(r["end_ms"]) for r in rows))
for r in tr: print("  ", r["kernel"], r["start_ms"], r["end_ms"], r["dur_ms"], r["grid"])
PY
