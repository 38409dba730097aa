#!/bin/bash
# Alignment-stage profile on the GPU box: the C4 bench with the consensus stage skipped (RVN_POLISH_SKIP_POA), once timed,
# once under rocprofv3 --pmc with the SQ counters; summary per kernel into gpurun_out/<tag>_nw_sq.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
export RVN_POLISH_SKIP_POA=1
export RVN_LIB_PATH=$R/raven_amd/lib/libraven_hip_test.so  # (the switches here exist in the debug build only)
RVN_NW_DEBUG=1 python $R/bench.py --no-cpu-baseline --steps 1 > $R/gpurun_out/${TAG}_nw_bench.json 2> $R/gpurun_out/${TAG}_nw_bench.err
grep "nw:" $R/gpurun_out/${TAG}_nw_bench.err | tail -2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/${TAG}_nw_pmc -o p -- python $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/${TAG}_nw_pmc.err
F=$(find $R/gpurun_out/${TAG}_nw_pmc -name '*counter_collection.csv' | head -1)
python $R/tools/pmc_summary.py "$F" $R/gpurun_out/${TAG}_nw_sq.csv
rm -rf $R/gpurun_out/${TAG}_nw_pmc
grep "nw_\|Kernel" $R/gpurun_out/${TAG}_nw_sq.csv | head -60
