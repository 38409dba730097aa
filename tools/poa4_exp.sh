#!/bin/bash
# Timing experiments on the NW step of poa4.hip (P4_EXP variants of the DEBUG library; results are wrong by design, only
# "dp cycles / dp wave-steps" is read).  build: cross-compiles the variants here;  run <tag>: on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/raven_amd/csrc
VARIANTS="${P4_VARIANTS:-0 1 2 3 4 6}"
case $1 in
  build)
    bash $C/build.sh > /dev/null
    mkdir -p $C/obj_exp $R/raven_amd/lib_exp
    for n in $VARIANTS; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DRVN_TEST_HOOKS -DRVN_DEBUG_KNOBS -DP4_EXP=$n -c $C/poa4.hip -o $C/obj_exp/poa4_$n.o &
    done
    wait
    for n in $VARIANTS; do
      objs=""
      for f in $C/obj_test/*.o; do [ "$(basename $f)" = "poa4.o" ] || objs="$objs $f"; done
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o $R/raven_amd/lib_exp/libraven_hip_test_$n.so $objs $C/obj_exp/poa4_$n.o -lz
    done
    ls -la $R/raven_amd/lib_exp/;;
  run)
    TAG=$2
    mkdir -p $R/gpurun_out
    for n in $VARIANTS; do
      RVN_LIB_PATH=$R/raven_amd/lib_exp/libraven_hip_test_$n.so RVN_POA_STATS=1 RVN_POA_MODES=9 timeout 300 python $R/tools/bench_poa.py ${N:-24576} 0 > $R/gpurun_out/${TAG}_exp$n.log 2> $R/gpurun_out/${TAG}_exp$n.err
      python - $n $R/gpurun_out/${TAG}_exp$n.log $R/gpurun_out/${TAG}_exp$n.err <<'PY'
import json, re, sys
n, log, err = sys.argv[1:]
d = [json.loads(l) for l in open(log) if l.startswith("{")]
m = re.search(r"dp wave-steps (\d+) \(dp cycles (\d+), in service points (\d+)\)", open(err).read())
if d and m:
    steps, cyc, sp = map(int, m.groups())
    print("exp", n, "device_ms", round(d[0]["device_ms"], 1), "dp cycles/step", round(cyc / max(steps, 1), 1), "of which service points", round(sp / max(steps, 1), 1),
          "steps", steps, "status", d[0]["status_counts"], "phase", {k: round(v / 1e9, 1) for k, v in d[0]["phase_cycles"].items()})
else:
    print("exp", n, "no result", open(err).read()[-300:])
PY
    done;;
esac
