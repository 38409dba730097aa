#!/bin/bash
# SQ counters of the window-consensus kernels (poa4_* and the poa2 fallback) on tools/bench_poa.py (two rocprofv3 --pmc passes); summary -> gpurun_out/<tag>_poa_sq.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
N=${2:-24576}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "poa4_[a-z]+_kernel|poa2_kernel" --output-format csv -d $R/gpurun_out/${TAG}_poa_pmc1 -o p -- python $R/tools/bench_poa.py $N 0 > /dev/null 2> $R/gpurun_out/${TAG}_poa_pmc1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-include-regex "poa4_[a-z]+_kernel|poa2_kernel" --output-format csv -d $R/gpurun_out/${TAG}_poa_pmc2 -o p -- python $R/tools/bench_poa.py $N 0 > /dev/null 2> $R/gpurun_out/${TAG}_poa_pmc2.err
for i in 1 2; do
  F=$(find $R/gpurun_out/${TAG}_poa_pmc$i -name '*counter_collection.csv' | head -1)
  python $R/tools/pmc_summary.py "$F" $R/gpurun_out/${TAG}_poa_sq$i.csv
  rm -rf $R/gpurun_out/${TAG}_poa_pmc$i
  cat $R/gpurun_out/${TAG}_poa_sq$i.csv
done
