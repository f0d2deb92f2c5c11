#!/bin/bash
# Build libnvq.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function ${NVQ_EXTRA_FLAGS:-}"
BUILD=build
OUT=libnvq.so
# NVQ_DEBUG_TOOLS=1: the diagnostic build for tools/ (nvq_debug_* entry points, tools/nvq_debug.h) as libnvq_debug.so; the
# shipped libnvq.so has neither those symbols nor any mutable state
if [ "${NVQ_DEBUG_TOOLS:-0}" = "1" ]; then FLAGS="$FLAGS -DNVQ_DEBUG_TOOLS"; BUILD=build_debug; OUT=libnvq_debug.so; fi
mkdir -p $BUILD
pids=()
# every object depends on every header (csrc/*.h are shared between translation units: a stale object would link
# mixed tile / pack layouts without any error)
newest_header=$(ls -t csrc/*.h ../include/*.h | head -1)
for f in csrc/*.hip; do
  o=$BUILD/$(basename "${f%.hip}").o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_header" -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $BUILD/*.o -o $OUT
echo "built $(pwd)/$OUT"
