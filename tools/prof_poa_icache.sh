#!/bin/bash
# instruction-cache / issue counters of the POA kernel on tools/bench_poa.py; summaries -> gpurun_out/<tag>_poa_ic*.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
N=${2:-12288}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "poa[234]_kernel" --output-format csv -d $R/gpurun_out/${TAG}_poa_ic$i -o p -- python $R/tools/bench_poa.py $N 0 > /dev/null 2> $R/gpurun_out/${TAG}_poa_ic$i.err
  F=$(find $R/gpurun_out/${TAG}_poa_ic$i -name '*counter_collection.csv' | head -1)
  if [ -n "$F" ]; then python $R/tools/pmc_summary.py "$F" $R/gpurun_out/${TAG}_poa_ic$i.csv > /dev/null; cat $R/gpurun_out/${TAG}_poa_ic$i.csv; else tail -3 $R/gpurun_out/${TAG}_poa_ic$i.err; fi
  rm -rf $R/gpurun_out/${TAG}_poa_ic$i
done
