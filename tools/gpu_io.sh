#!/bin/bash
# input path on the GPU box: parity tests + the stand-alone timing
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_io.py -x -q 2>&1 | tail -15 > gpurun_out/io_tests.log
cat gpurun_out/io_tests.log
timeout 300 python tools/bench_io.py 150 > gpurun_out/io_bench.json 2> gpurun_out/io_bench.err
cat gpurun_out/io_bench.json; tail -3 gpurun_out/io_bench.err
RVN_IO_THREADS=64 timeout 300 python tools/bench_io.py 150 > gpurun_out/io_bench_t64.json 2>> gpurun_out/io_bench.err
cat gpurun_out/io_bench_t64.json
