#!/bin/bash
# Build libnvq.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function"
mkdir -p build
pids=()
for f in csrc/*.hip; do
  o=build/$(basename "${f%.hip}").o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ csrc/common.h -nt "$o" ] || [ ../include/nvq.h -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -o libnvq.so
echo "built $(pwd)/libnvq.so"
