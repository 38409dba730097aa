# round r + 1's targets from HBM (default) against an upload of what round r returned (--targets-through-host), alternating: C4 / C2
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); lp = d['last_polish_round']
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'overlap_s', d['legs']['overlap_s_of_each_step'], 'polish_s', d['legs']['polish_s_of_each_step'], 'round total_ms', lp['total_ms'])"; }
for rep in 1 2 3; do
  timeout 600 python bench.py --workload ${W:-c4} --steps 2 --warmup 1 --no-cpu-baseline --load-bases 0 2>/dev/null | line "resident    "
  timeout 600 python bench.py --workload ${W:-c4} --steps 2 --warmup 1 --no-cpu-baseline --load-bases 0 --targets-through-host 2>/dev/null | line "through host"
done
