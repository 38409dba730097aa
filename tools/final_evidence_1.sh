#!/bin/bash
# round evidence, part 1: consensus parity on 20 000 windows, the default bench line (C4) and the C2 line
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
mkdir -p $R/gpurun_out
cd $R
timeout 600 python tools/poa_parity.py 20000 > gpurun_out/${TAG}_poa_parity_20000.json 2> gpurun_out/${TAG}_poa_parity.err
cut -c1-600 gpurun_out/${TAG}_poa_parity_20000.json
timeout 900 python bench.py > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err
tail -1 gpurun_out/${TAG}_bench_c4.json | cut -c1-400
timeout 600 python bench.py --workload c2 > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err
tail -1 gpurun_out/${TAG}_bench_c2.json | cut -c1-400
