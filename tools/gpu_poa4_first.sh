#!/bin/bash
# first GPU run of poa4: identity vs poa2 on ragged windows, throughput of mode 9 vs mode 2, SQ counters
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 300 python tools/check_poa3.py 400 9 > gpurun_out/r04_check_poa4.log 2>&1
echo "check rc $?" >> gpurun_out/r04_check_poa4.log
RVN_POA_MODES=9,2 timeout 900 python tools/bench_poa.py ${1:-20000} 40 > gpurun_out/r04_bench_poa_first.json 2> gpurun_out/r04_bench_poa_first.err
RVN_POA_MODE=9 timeout 600 bash tools/prof_poa.sh r04a 6000 > gpurun_out/r04a_prof.log 2>&1
tail -5 gpurun_out/r04_check_poa4.log
cat gpurun_out/r04_bench_poa_first.json
