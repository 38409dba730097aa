#!/bin/bash
# Cache / memory-pipe counters of the window-consensus kernel on tools/bench_poa.py (one rocprofv3 --pmc pass per group);
# summary -> gpurun_out/<tag>_poa_mem<i>.csv.  The counter names available on the box go to gpurun_out/<tag>_counters.txt.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
N=${2:-24576}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCC|TCP|TA|TD|SQ|GRBM)_[A-Za-z0-9_]+" | sort -u > $R/gpurun_out/${TAG}_counters.txt
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "poa4_persistent_kernel" --output-format csv -d $R/gpurun_out/${TAG}_poa_mem$i -o p -- python $R/tools/bench_poa.py $N 0 > /dev/null 2> $R/gpurun_out/${TAG}_poa_mem$i.err
  F=$(find $R/gpurun_out/${TAG}_poa_mem$i -name '*counter_collection.csv' | head -1)
  if [ -n "$F" ]; then python $R/tools/pmc_summary.py "$F" $R/gpurun_out/${TAG}_poa_mem$i.csv; cat $R/gpurun_out/${TAG}_poa_mem$i.csv; else tail -3 $R/gpurun_out/${TAG}_poa_mem$i.err; fi
  rm -rf $R/gpurun_out/${TAG}_poa_mem$i
done
