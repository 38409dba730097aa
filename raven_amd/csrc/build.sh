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
for f in scan radix_sort sketch index map pile edit_distance poa poa2 poa4 simt_emu polish nwpath pass2 io shard group engine $EDLIB; do
  src="$HERE/$f.hip"; obj="$HERE/obj/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ -n "$(find "$HERE" -maxdepth 1 -name '*.h' -newer "$obj" -print -quit)" ] || [ "$HERE/../../include/raven_hip.h" -nt "$obj" ]; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libraven_hip.so" "$HERE"/obj/*.o -lz
echo "built $OUT/libraven_hip.so"
