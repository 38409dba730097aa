#!/bin/bash
# Collects the evidence files of a round on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (the N = 1 line: 100 Mb, -p 2)
#   2. / 3. separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of the same command (HBM traffic per launch)
# Outputs under gpurun_out/<tag>_*; tools/pmc_traffic.py + the stats CSV are what gets copied to profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
WORKLOAD=${2:-c4}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o s -- python $R/bench.py --workload $WORKLOAD --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o p -- python $R/bench.py --workload $WORKLOAD --no-cpu-baseline --no-kernel-timing > /dev/null 2> $R/gpurun_out/${TAG}_pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o p -- python $R/bench.py --workload $WORKLOAD --no-cpu-baseline --no-kernel-timing > /dev/null 2> $R/gpurun_out/${TAG}_pmc_write.err
F=$(find $R/gpurun_out/${TAG}_pmc_fetch -name '*counter_collection.csv' | head -1)
W=$(find $R/gpurun_out/${TAG}_pmc_write -name '*counter_collection.csv' | head -1)
# the default command = (2 warm-up + 2 timed) steps x 2 polishing rounds
# what the counters report for known byte counts on this box (streams, gathers, scattered stores)
bash $R/tools/pmc_calibrate.sh $R/gpurun_out/${TAG}_pmc_calibration.json > $R/gpurun_out/${TAG}_pmc_calibration.log 2>&1
python $R/tools/pmc_traffic.py "$F" "$W" $R/gpurun_out/${TAG}_pmc_traffic.json 8 $R/gpurun_out/${TAG}_pmc_calibration.json
S=$(find $R/gpurun_out/${TAG}_stats -name '*kernel_stats.csv' | head -1)
cp "$S" $R/gpurun_out/${TAG}_kernel_stats.csv
# the raw per-dispatch files are large: keep only the summaries
rm -rf $R/gpurun_out/${TAG}_pmc_fetch $R/gpurun_out/${TAG}_pmc_write $R/gpurun_out/${TAG}_stats
head -12 $R/gpurun_out/${TAG}_kernel_stats.csv
tail -1 $R/gpurun_out/${TAG}_bench_under_rocprof.json | cut -c1-300
