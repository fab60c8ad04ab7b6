#!/bin/bash
# Builds libjnroll.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${JN_LIB_OUT:-$HERE/../lib}"
mkdir -p "$OUT" "$HERE/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result ${JN_EXTRA_FLAGS:-}"
pids=()
for f in kernels_conv kernels_pwres kernels_pwxs kernels_bwd kernels_train kernels_gptbwd kernels_det kernels_detloss kernels_aug kernels_env kernels_gpt kernels_gptmfma api; do
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ -n "$(find "$HERE" -maxdepth 1 -name '*.h' -newer "$HERE/obj/$f.o")" ] || [ "$HERE/../../include/jnroll.h" -nt "$HERE/obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
$HIPCC $FLAGS -x hip -c "$HERE/plan.cpp" -o "$HERE/obj/plan.o" &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC -shared -fPIC --offload-arch=gfx950 "$HERE"/obj/*.o -o "$OUT/libjnroll.so"
echo "built $OUT/libjnroll.so"
