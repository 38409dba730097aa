#!/bin/bash
# A/B of environment settings on the C4 bench line (one warm-up + one timed step each): usage gpu_ab_bench.sh "VAR=a" "VAR=b" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for SET in "$@"; do
  env $SET timeout 400 python bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-timing --load-bases 0 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
lp=d["last_polish_round"]
print("$SET", "value", d["value"], "step_ms", d["ms_per_step"], "polish_s", d["legs"]["polish_s_of_each_step"], "poa_ms", lp["poa_ms"], "map", lp["map_ms"], "align", lp["align_ms"])
PY
done
