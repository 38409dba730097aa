#!/bin/bash
# chain stage: bit-exact parity tests (map / pass-2 / polish stages incl. the HiFi workload), then the HiFi bench step
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_pass2.py tests/test_gpu_polish.py -x -q 2>&1 | tail -8 > gpurun_out/chain_tests.log
cat gpurun_out/chain_tests.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "configs4 or configs2" 2>&1 | tail -5 >> gpurun_out/chain_tests.log
tail -5 gpurun_out/chain_tests.log
timeout 600 python bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --load-bases 0 > gpurun_out/chain_c5.json 2> gpurun_out/chain_c5.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/chain_c5.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], d["legs"]["overlap_s_per_step"], d["legs"]["polish_s_per_step"], d["c5_stages"]["seconds_per_step"])
for k in ("chain","seg_sort_group","seg_sort_pos","edit_lane","match_emit"):
    print(k, d["kernels"][k]["ms_per_step"])
PY
