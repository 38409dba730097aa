#!/bin/bash
# round evidence, part 4 (after the arena and the buffer-rotation fix): the HiFi line and the default line again
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python bench.py --workload c5 --steps 2 --warmup 2 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err
tail -1 gpurun_out/${TAG}_bench_c5.json | cut -c1-200
timeout 900 python bench.py > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err
tail -1 gpurun_out/${TAG}_bench_c4.json | cut -c1-200
