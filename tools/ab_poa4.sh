cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in lib_b lib_a; do
  RVN_LIB_PATH=$PWD/raven_amd/$lib/libraven_hip_test.so RVN_POA_REPS=3 RVN_POA_MODES="9" timeout 300 python tools/bench_poa.py 24576 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', 'mode', d['run'], 'device_ms', round(d['device_ms'], 1), {k: round(v / 1e9, 1) for k, v in d['phase_cycles'].items()})"
done
done
for lib in lib_b lib_a; do
  RVN_LIB_PATH=$PWD/raven_amd/$lib/libraven_hip_test.so timeout 600 python bench.py --workload c4 --steps 2 --warmup 2 --no-cpu-baseline --load-bases 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$lib', 'ms/step', d['ms_per_step'], 'poa_ms', d['last_polish_round']['poa_ms'], 'poa_rows avg', k['poa_rows']['avg_launch_ms'], 'banded', k.get('poa_banded',{}).get('ms_per_step'), 'to64', d['last_polish_round']['poa_windows_to_64_columns'])"
done
