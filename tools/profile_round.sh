#!/bin/bash
# Collects the evidence files of a round on the GPU box (run through gpurun): POA bench, PMC passes of the banded
# POA kernel, rocprofv3 kernel stats of bench.py.  Outputs under gpurun_out/; copy what should be judged to profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01_i}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
RVN_POA_MODE=2 python $R/tools/bench_poa.py 16384 100 > $R/gpurun_out/${TAG}_poa_banded_bench.json 2>/dev/null
RVN_POA_MODE=2 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/${TAG}_pmc_a -o p -- python $R/tools/bench_poa.py 16384 0 > /dev/null 2>&1
RVN_POA_MODE=2 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/${TAG}_pmc_b -o p -- python $R/tools/bench_poa.py 16384 0 > /dev/null 2>&1
RVN_POA_MODE=2 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o p -- python $R/tools/bench_poa.py 16384 0 > /dev/null 2>&1
RVN_POA_MODE=2 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o p -- python $R/tools/bench_poa.py 16384 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
ls $R/gpurun_out/${TAG}_stats | head
