# Window-consensus stage on small batches: first attempt alone (mode 9), the 64-column kernel (mode 2) and
# the default chain (mode 0) on n unguided windows; RVN_POA_GW = windows per group of the first attempt (debug build).
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() {  # n modes
  RVN_LIB_PATH=$PWD/raven_amd/lib/libraven_hip_test.so RVN_POA_REPS=3 RVN_POA_MODES="$2" timeout 300 python tools/bench_poa.py $1 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n', $1, 'mode', d.get('run', d.get('mode')), 'device_ms', round(d['device_ms'], 1), 'windows/s', round(d.get('windows_per_s', 0)), 'status', d.get('status_counts'))"
}
run 2500 "9@RVN_POA_GW=4,9@RVN_POA_GW=2,9@RVN_POA_GW=1,2"
run 5000 "9@RVN_POA_GW=4,9@RVN_POA_GW=2,9,2"
run 10000 "9@RVN_POA_GW=4,9@RVN_POA_GW=3,9,2,0"
run 16384 "9,2,0"
run 25000 "9,2,0"
