#!/bin/bash
# round evidence, part 3: SQ counters of the window-consensus kernels; the HiFi (configs[4] workload) bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
mkdir -p $R/gpurun_out
cd $R
bash tools/prof_poa.sh $TAG 12288 2>&1 | tail -30
cd $R
timeout 900 python bench.py --workload c5 --steps 2 --warmup 2 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err
tail -1 gpurun_out/${TAG}_bench_c5.json | cut -c1-300
