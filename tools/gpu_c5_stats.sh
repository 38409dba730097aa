#!/bin/bash
# HiFi workload (configs[4] on one GPU): kernel stats of one warm-up + one timed step
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_c5_stats -o s -- python $R/bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --load-bases 0 > $R/gpurun_out/r04_c5_bench_under_rocprof.json 2> $R/gpurun_out/r04_c5_stats.err
S=$(find $R/gpurun_out/r04_c5_stats -name '*kernel_stats.csv' | head -1)
cp "$S" $R/gpurun_out/r04_c5_kernel_stats.csv
rm -rf $R/gpurun_out/r04_c5_stats
head -25 $R/gpurun_out/r04_c5_kernel_stats.csv | cut -c1-200
tail -1 $R/gpurun_out/r04_c5_bench_under_rocprof.json | cut -c1-1500
