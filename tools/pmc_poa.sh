R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export RVN_POA_MODE=2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_poa_a -o p -- python $R/tools/bench_poa.py 12288 0 > $R/gpurun_out/pmc_poa_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_poa_b -o p -- python $R/tools/bench_poa.py 12288 0 > $R/gpurun_out/pmc_poa_b.log 2>&1
ls -R $R/gpurun_out/pmc_poa_a | head
