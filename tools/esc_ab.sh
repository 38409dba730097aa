# same binary, the in-launch 64-column queue on / off (RVN_POA_NO_ESC, debug build), alternating: C4
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']; lp = d['last_polish_round']
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'polish_s', d['legs']['polish_s_of_each_step'], 'poa_ms', lp['poa_ms'], 'poa_rows avg', k.get('poa_rows',{}).get('avg_launch_ms'), 'banded ms/step', k.get('poa_banded',{}).get('ms_per_step'), 'to64', lp['poa_windows_to_64_columns'])"; }
for rep in 1 2 3; do
  RVN_LIB_PATH=$PWD/raven_amd/lib/libraven_hip_test.so timeout 600 python bench.py --workload ${W:-c4} --steps 2 --warmup 1 --no-cpu-baseline --load-bases 0 $EXTRA 2>/dev/null | line "queue   "
  RVN_POA_NO_ESC=1 RVN_LIB_PATH=$PWD/raven_amd/lib/libraven_hip_test.so timeout 600 python bench.py --workload ${W:-c4} --steps 2 --warmup 1 --no-cpu-baseline --load-bases 0 $EXTRA 2>/dev/null | line "no queue"
done
