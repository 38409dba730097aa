#!/bin/bash
# A/B of two builds of the library on tools/bench_poa.py: raven_amd/lib_a/libraven_hip.so (A) against raven_amd/lib (B),
# alternating, three timed runs per process.  usage: gpu_ab_lib.sh [windows] [modes]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-24576}
M=${2:-9,9,9}
cd $R
for rep in 1 2; do
  for L in lib_a lib; do
    RVN_LIB_PATH=$R/raven_amd/$L/libraven_hip.so RVN_POA_MODES=$M timeout 300 python tools/bench_poa.py $N 0 2>/dev/null | python -c "
import sys, json
print('$L', [round(json.loads(l)['device_ms'], 1) for l in sys.stdin if l.startswith('{')])"
  done
done
