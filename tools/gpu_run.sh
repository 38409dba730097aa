#!/bin/bash
# One gpurun call of round 5: runs the named steps on the GPU box, everything under its own timeout, outputs under
# gpurun_out/<tag>_*.  usage: gpu_run.sh <tag> step [step ...]
#   tests-poa | tests-mgpu | tests-all | smoke | benchpoa:<modes> | c4[:ENV=V,...] | c2[:ENV=V,...] | c5 | bench-full:<c4|c2|c5>
#   | parity:<n> | profile:<tag> | sqpoa:<tag> | sqnw:<tag> | tracenw:<tag>       (ENV=V switches need the debug build: the
#   script points RVN_LIB_PATH at libraven_hip_test.so for those runs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
TAG=$1; shift
summ() {  # one compact line per bench JSON
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print(sys.argv[1], "no JSON line:", e); sys.exit(0)
lp = d.get("last_polish_round") or {}
print(sys.argv[1].split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "overlap_s", d["legs"]["overlap_s_per_step"],
      "polish_s", d["legs"]["polish_s_per_step"], "poa_ms", lp.get("poa_ms"), "map_ms", lp.get("map_ms"), "align_ms", lp.get("align_ms"),
      "windows/s", d["legs"].get("windows_per_s"), "frac", (d.get("roofline") or {}).get("frac"),
      "frac_computed", (d.get("roofline") or {}).get("frac_on_computed_cells"))
PY
}
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  envs=""; [ -n "$arg" ] && envs=$(echo "$arg" | tr ',' ' ')
  t0=$(date +%s)
  case $name in
    tests-poa) timeout 900 python -m pytest tests/test_gpu_poa.py tests/test_gpu_polish.py -x -q -m gpu > gpurun_out/${TAG}_tests_poa.log 2>&1; tail -4 gpurun_out/${TAG}_tests_poa.log;;
    tests-mgpu) timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_group.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/${TAG}_tests_mgpu.log 2>&1; tail -6 gpurun_out/${TAG}_tests_mgpu.log;;
    tests-all) timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests_all.log 2>&1; tail -4 gpurun_out/${TAG}_tests_all.log;;
    benchpoa_a) RVN_LIB_PATH=$R/raven_amd/lib_a/libraven_hip_test.so RVN_POA_MODES="$arg" timeout 900 python tools/bench_poa.py 24576 0 2>gpurun_out/${TAG}_benchpoa_a.err | tee gpurun_out/${TAG}_benchpoa_a.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('benchpoa lib_a', d.get('run', d.get('mode')), 'device_ms', round(d['device_ms'], 1), 'windows/s', round(d.get('windows_per_s', 0)))";;
    benchpoa) RVN_LIB_PATH=$R/raven_amd/lib/libraven_hip_test.so RVN_POA_MODES="$arg" timeout 900 python tools/bench_poa.py 24576 0 2>gpurun_out/${TAG}_benchpoa.err | tee gpurun_out/${TAG}_benchpoa.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('benchpoa', d.get('run', d.get('mode')), 'device_ms', round(d['device_ms'], 1), 'windows/s', round(d.get('windows_per_s', 0)))";;
    c4_a) f=gpurun_out/${TAG}_c4a_$(echo "$arg" | tr -c 'A-Za-z0-9\n' '_').json
       env RVN_LIB_PATH=$R/raven_amd/lib_a/libraven_hip_test.so $envs timeout 900 python bench.py --workload c4 --steps ${STEPS:-2} --warmup ${WARMUP:-2} --no-cpu-baseline --load-bases 0 > $f 2> ${f%.json}.err; summ $f;;
    c4|c2|c5) f=gpurun_out/${TAG}_${name}_$(echo "$arg" | tr -c 'A-Za-z0-9\n' '_').json
       [ -n "$envs" ] && envs="RVN_LIB_PATH=$R/raven_amd/lib/libraven_hip_test.so $envs"  # (environment switches exist in the debug build only)
       env $envs timeout 900 python bench.py --workload $name --steps ${STEPS:-2} --warmup ${WARMUP:-2} --no-cpu-baseline --load-bases 0 $BENCH_ARGS > $f 2> ${f%.json}.err; summ $f;;
    bench-full) f=gpurun_out/${TAG}_bench_${arg}.json   # the line the driver takes (default steps; CPU baseline at c4)
       extra="--no-cpu-baseline"; [ "$arg" = "c4" ] && extra=""
       timeout 1500 python bench.py --workload $arg $extra > $f 2> ${f%.json}.err; summ $f;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3;;
    parity) n=${arg%%/*}; shape=ont; [ "$arg" != "$n" ] && shape=${arg#*/}   # parity:20000 or parity:20000/qual
       f=gpurun_out/${TAG}_poa_parity_${shape}_$n.json
       timeout 1200 python tools/poa_parity.py $n 0 0 $shape > $f 2> gpurun_out/${TAG}_poa_parity.err; python -c "
import json; d = json.load(open('$f')); print({k: v for k, v in d.items() if k not in ('examples', 'not_explained')}, 'unexplained', len(d['not_explained']))";;
    polishparity) n=${arg%%/*}; shape=q10; [ "$arg" != "$n" ] && shape=${arg#*/}   # polishparity:200/q10 (contigs of 100 windows)
       f=gpurun_out/${TAG}_polish_parity_${shape}_$n.json
       timeout 1500 python tools/polish_parity.py $n 0 $shape > $f 2> gpurun_out/${TAG}_polish_parity.err; python -c "
import json; d = json.load(open('$f')); print({k: v for k, v in d.items() if k != 'different'})";;
    rankshare) f=gpurun_out/${TAG}_rank_share_$(echo "$arg" | tr -c 'A-Za-z0-9\n' '_').json   # rankshare:--ranks=8,--workload=c4
       timeout 1200 python tools/rank_share.py $(echo "$arg" | tr ',' ' ') > $f 2> ${f%.json}.err; python -c "
import json; d = json.loads([l for l in open('$f') if l.startswith('{')][-1])
print({k: d[k] for k in ('ranks', 'single_gpu', 'predicted_step_s_at_n_ranks', 'ideal_step_s', 'predicted_strong_scaling_efficiency', 'sharded_rounds_equal_single_gpu_consensus')})
print('pass', {k: v for k, v in d['overlap_pass_at_n_ranks'].items() if k != 'stats_rank0'})
for r in d['polishing_rounds_at_n_ranks']: print(r)" || tail -5 ${f%.json}.err;;
    profile) bash tools/profile_round.sh $arg
       # (a bench-full step later in the same call reads the traffic file from profiles/: without this its line says stale)
       for f in pmc_traffic.json pmc_calibration.json kernel_stats.csv bench_under_rocprof.json; do [ -f gpurun_out/${arg}_$f ] && cp gpurun_out/${arg}_$f profiles/; done;;
    sqpoa) bash tools/prof_poa.sh $arg;;
    mempoa) bash tools/prof_poa_mem.sh $arg;;
    tracenw) bash tools/trace_nw.sh $arg;;
    sqnw) bash tools/prof_nw.sh $arg;;
    *) echo "unknown step $step";;
  esac
  echo "[$step] $(( $(date +%s) - t0 )) s"
done
