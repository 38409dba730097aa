# same-box A/B of product libraries: bash tools/ab_lib.sh lib_a lib lib_X ...   (directories under raven_amd/)
for lib in "$@"; do
  RVN_LIB_PATH=$PWD/raven_amd/$lib/libraven_hip.so timeout 600 python bench.py --workload ${WORKLOAD:-c4} --steps 2 --warmup 2 --no-cpu-baseline --load-bases 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$lib', 'ms/step', d['ms_per_step'], 'poa_ms', d['last_polish_round']['poa_ms'], 'poa_rows avg', k['poa_rows']['avg_launch_ms'], 'map_ms', d['last_polish_round']['map_ms'], 'align_ms', d['last_polish_round']['align_ms'])"
done
