#!/bin/bash
# tools/fetch_real_deps.sh — OPT-IN, needs a network (there is none in the build container, so this script has never
# run there; it is the recipe SURVEY.md §7 asks for).  Fetches the dependencies the reference pulls through CMake
# FetchContent (Raven.deps.cmake:43-44 -> lbcb-sci/racon @ library, which itself fetches ram, spoa, edlib, biosoup,
# thread_pool, bioparser), builds them out of tree under oracle/_ref/deps, builds tools/real_deps/ref_harness.cpp
# against them and diffs the real libraries' output with oracle/ on the reference's lambda data:
#   ram    Minimize / Filter(0.001) / Map(avoid_equal, avoid_symmetric[, minhash]) per read  -> overlap lists, occurrence
#   edlib  NW edit distance on read pairs
#   racon  one Polish round of the lambda reads on the lambda genome                         -> consensus
# A clean diff is what turns "parity unpinned" (oracle/raven_oracle.cpp header) into "pinned".
# Nothing fetched or built here is committed: oracle/_ref/ is git-ignored.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
DEPS="$ROOT/oracle/_ref/deps"
mkdir -p "$DEPS"
command -v git >/dev/null && command -v cmake >/dev/null || { echo "needs git and cmake" >&2; exit 2; }
if ! git ls-remote https://github.com/lbcb-sci/racon >/dev/null 2>&1; then
  echo "no network: cannot reach github.com (this script is opt-in; the repo's tests do not depend on it)" >&2
  exit 3
fi
[ -d "$DEPS/racon" ] || git clone --depth 1 --branch library https://github.com/lbcb-sci/racon "$DEPS/racon"
cmake -S "$DEPS/racon" -B "$DEPS/build" -DCMAKE_BUILD_TYPE=Release -Dracon_build_tests=OFF -Dracon_build_exe=OFF
cmake --build "$DEPS/build" -j "$(nproc)"
# the fetched sources sit under build/_deps/<name>-src, the static libraries under build/ (names as of the pinned tags)
INC=""
for d in "$DEPS/racon/include" "$DEPS"/build/_deps/*-src/include "$DEPS"/build/_deps/edlib-src/edlib/include; do
  [ -d "$d" ] && INC="$INC -I$d"
done
LIBS="$(find "$DEPS/build" -name '*.a' | tr '\n' ' ')"
g++ -O2 -std=c++17 -pthread $INC "$ROOT/tools/real_deps/ref_harness.cpp" -Wl,--start-group $LIBS -Wl,--end-group -lz \
    -o "$ROOT/oracle/_ref/ref_harness"
python3 "$ROOT/tools/real_deps/compare_with_oracle.py" "$ROOT/oracle/_ref/ref_harness"
