# The 64-column attempt inside the persistent launch (poa4.hip, poa4_esc_*): tests, C4 A/B against the build in raven_amd/lib_a,
# small batches with the first attempt from 0 windows up, C2 with and without the 20 000-window threshold.
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_poa.py tests/test_gpu_polish.py -x -q -m gpu 2>&1 | tail -3
line() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']; lp = d['last_polish_round']
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'poa_ms', lp['poa_ms'], 'poa_rows avg', k.get('poa_rows',{}).get('avg_launch_ms'), 'banded ms/step', k.get('poa_banded',{}).get('ms_per_step'), 'to64', lp['poa_windows_to_64_columns'], 'to128', lp['poa_windows_to_128_columns'])"; }
for lib in lib_b lib_a; do
  RVN_LIB_PATH=$PWD/raven_amd/$lib/libraven_hip_test.so timeout 600 python bench.py --workload c4 --steps 2 --warmup 2 --no-cpu-baseline --load-bases 0 2>/dev/null | line "c4 $lib"
done
RVN_POA_NO_ESC=1 RVN_LIB_PATH=$PWD/raven_amd/lib_b/libraven_hip_test.so timeout 600 python bench.py --workload c4 --steps 2 --warmup 2 --no-cpu-baseline --load-bases 0 2>/dev/null | line "c4 lib_b no queue"
timeout 600 python bench.py --workload c2 --steps 3 --warmup 2 --no-cpu-baseline --load-bases 0 2>/dev/null | line "c2 default"
timeout 600 python bench.py --workload c2 --steps 3 --warmup 2 --no-cpu-baseline --load-bases 0 --engine-option poa_rows_min_windows=0 2>/dev/null | line "c2 min_windows=0"
for n in 1000 2500 5000 10000 25000; do
  RVN_LIB_PATH=$PWD/raven_amd/lib/libraven_hip_test.so RVN_POA_REPS=3 RVN_POA_MODES="9,2,0@RVN_POA_MIN=0" timeout 300 python tools/bench_poa.py $n 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n', $n, 'mode', d.get('run', d.get('mode')), 'device_ms', round(d['device_ms'], 1), 'status', d.get('status_counts'))"
done
