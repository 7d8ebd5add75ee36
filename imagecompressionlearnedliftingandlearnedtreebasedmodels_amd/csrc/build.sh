#!/bin/bash
# Builds liblldwt.so (gfx950 only) in-tree, next to the Python package.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../liblldwt.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable"
mkdir -p "$HERE/obj"
pids=()
for f in ops lifting lifting_f16 cdf97 conv_mfma conv_f16x3 conv_wgrad_f16x3 cgp_fused cgp_f16x3 conv_bwd rans; do
  [ -f "$HERE/$f.hip" ] || continue
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ "$HERE/common.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/lifting_f16.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/split_f16.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/../../include/lldwt.h" -nt "$HERE/obj/$f.o" ]; then
    EXTRA=""
    # the 36-unit chunk loop of the split-fp16 conv must unroll completely (register rings indexed by the unit number)
    [ "$f" = conv_f16x3 ] && EXTRA="-mllvm -pragma-unroll-threshold=131072"
    ( $HIPCC $FLAGS $EXTRA -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/obj/*.o
echo "built $OUT"
