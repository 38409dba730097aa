#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
RVN_DEBUG_MEM=1 timeout 600 python bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --load-bases 0 > gpurun_out/c5_mem.json 2> gpurun_out/c5_mem.err
grep -c "DevBuf grows" gpurun_out/c5_mem.err
grep "raven_hip" gpurun_out/c5_mem.err | tail -150
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c5_mem.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], d["legs"]["overlap_s_per_step"], d["legs"]["polish_s_per_step"], d["c5_stages"]["seconds_per_step"])
PY
