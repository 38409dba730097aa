#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
RVN_DEBUG_MEM=1 timeout 700 python bench.py --workload c5 --steps 2 --warmup 2 --no-cpu-baseline --load-bases 0 > gpurun_out/pool_c5.json 2> gpurun_out/pool_c5.err
grep -c "driver" gpurun_out/pool_c5.err; grep -c "ms, pool" gpurun_out/pool_c5.err
grep "stage repeated\|flushing" gpurun_out/pool_c5.err | tail
grep "stage entry" gpurun_out/pool_c5.err | tail -16
tail -3 gpurun_out/pool_c5.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/pool_c5.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], d["legs"]["overlap_s_of_each_step"], d["legs"]["polish_s_of_each_step"], d["c5_stages"]["seconds_per_step"], "poa round ms", d["roofline"]["avg_launch_ms"])
PY
