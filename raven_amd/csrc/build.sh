#!/bin/bash
# Builds libraven_hip.so for gfx950 in-tree (raven_amd/lib/). hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=()
# RVN_NO_EDLIB_SYMBOLS=1: leave the edlibAlign drop-in out (a process that also loads a real shared edlib: INTEGRATION.md 3.1)
EDLIB=edlib_dropin
if [ -n "${RVN_NO_EDLIB_SYMBOLS:-}" ]; then EDLIB=""; rm -f "$HERE/obj/edlib_dropin.o"; fi
PRODUCT="scan radix_sort sketch index map pile edit_distance poa poa2 poa4 polish nwpath pass2 io shard group engine $EDLIB"
# libraven_hip_test.so (TEST INFRASTRUCTURE, include/raven_hip_test.h): every source compiled again under
# -DRVN_TEST_HOOKS (rvn_test_*, rvn_poa_banded_emulate) -DRVN_DEBUG_KNOBS (the environment switches of experiments and
# diagnostics: common.h knob() — the product library reads none) + the host wavefront emulator
HOOKED="$PRODUCT"
mkdir -p "$HERE/obj_test"
stale() {  # $1 = source, $2 = object
  [ ! -f "$2" ] || [ "$1" -nt "$2" ] || [ -n "$(find "$HERE" -maxdepth 1 -name '*.h' -newer "$2" -print -quit)" ] || [ "$HERE/../../include/raven_hip.h" -nt "$2" ] || [ "$HERE/../../include/raven_hip_test.h" -nt "$2" ]
}
rm -f "$HERE/obj/simt_emu.o"
for f in $PRODUCT; do
  src="$HERE/$f.hip"; obj="$HERE/obj/$f.o"
  if stale "$src" "$obj"; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for f in $HOOKED simt_emu; do
  src="$HERE/$f.hip"; obj="$HERE/obj_test/$f.o"
  if stale "$src" "$obj"; then
    $HIPCC $FLAGS -DRVN_TEST_HOOKS -DRVN_DEBUG_KNOBS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
objs=(); tobjs=("$HERE/obj_test/simt_emu.o")
for f in $PRODUCT; do
  objs+=("$HERE/obj/$f.o")
  case " $HOOKED " in *" $f "*) tobjs+=("$HERE/obj_test/$f.o");; *) tobjs+=("$HERE/obj/$f.o");; esac
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o "$OUT/libraven_hip.so" "${objs[@]}" -lz
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o "$OUT/libraven_hip_test.so" "${tobjs[@]}" -lz
echo "built $OUT/libraven_hip.so $OUT/libraven_hip_test.so"
