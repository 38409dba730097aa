#!/bin/bash
# memory-side counters of the POA kernel on tools/bench_poa.py (separate rocprofv3 --pmc passes); summaries -> gpurun_out/<tag>_poa_mem*.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
N=${2:-12288}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "poa[234]_kernel" --output-format csv -d $R/gpurun_out/${TAG}_poa_mem$i -o p -- python $R/tools/bench_poa.py $N 0 > /dev/null 2> $R/gpurun_out/${TAG}_poa_mem$i.err
  F=$(find $R/gpurun_out/${TAG}_poa_mem$i -name '*counter_collection.csv' | head -1)
  if [ -n "$F" ]; then python $R/tools/pmc_summary.py "$F" $R/gpurun_out/${TAG}_poa_mem$i.csv; cat $R/gpurun_out/${TAG}_poa_mem$i.csv; else tail -3 $R/gpurun_out/${TAG}_poa_mem$i.err; fi
  rm -rf $R/gpurun_out/${TAG}_poa_mem$i
done
