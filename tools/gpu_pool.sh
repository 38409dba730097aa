#!/bin/bash
# the device arena behind the grow-only buffers: parity tests with every stage entry releasing its scratch into a small
# arena, then the HiFi bench line with the allocator trace
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
RVN_RELEASE_ALWAYS=1 RVN_ARENA_MB=24000 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pass2.py tests/test_gpu_polish.py tests/test_gpu_stages.py tests/test_gpu_sharded.py tests/test_gpu_group.py -x -q 2>&1 | tail -4
RVN_DEBUG_MEM=1 timeout 700 python bench.py --workload c5 --steps 2 --warmup 2 --no-cpu-baseline --load-bases 0 > gpurun_out/pool_c5.json 2> gpurun_out/pool_c5.err
grep -c "ms, driver" gpurun_out/pool_c5.err; grep -c "ms, arena" gpurun_out/pool_c5.err
grep "stage repeated\|arena of" gpurun_out/pool_c5.err | tail
grep "ms, driver" gpurun_out/pool_c5.err | awk '{ for (i=1;i<=NF;i++) if ($i ~ /^\(/) { v=substr($i,2)+0; if (v > 50) print } }' | tail -12
grep "stage entry" gpurun_out/pool_c5.err | tail -16
tail -3 gpurun_out/pool_c5.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/pool_c5.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], d["legs"]["overlap_s_of_each_step"], d["legs"]["polish_s_of_each_step"], d["c5_stages"]["seconds_per_step"], "poa round ms", d["roofline"]["avg_launch_ms"])
PY
