#!/bin/bash
# Builds liblldwt.so (gfx950 only) in-tree, next to the Python package.  hipcc cross-compiles without a GPU.
# Incremental on CONTENT, not on mtimes: an object is reused only if the sha1 of its source, the shared headers and the
# compile flags equals the key stored beside it (a checkout that resets mtimes cannot link stale objects).
# LLDWT_FORCE_BUILD=1 (or `build.sh --force`) recompiles everything.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../liblldwt.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable"
FORCE="${LLDWT_FORCE_BUILD:-0}"
[ "$1" = "--force" ] && FORCE=1
mkdir -p "$HERE/obj"
HDRS="$HERE/common.h $HERE/lifting_f16.h $HERE/split_f16.h $HERE/../../include/lldwt.h"
pids=()
names=()
for f in ops lifting lifting_f16 cdf97 conv_mfma conv_f16x3 conv_wgrad_f16x3 cgp_fused cgp_f16x3 conv_bwd rans; do
  [ -f "$HERE/$f.hip" ] || continue
  EXTRA=""
  # the 36-unit chunk loop of the split-fp16 conv must unroll completely (register rings indexed by the unit number)
  [ "$f" = conv_f16x3 ] && EXTRA="-mllvm -pragma-unroll-threshold=131072"
  key="$( (cat "$HERE/$f.hip" $HDRS; echo "$HIPCC $FLAGS $EXTRA") | sha1sum | cut -d' ' -f1)"
  if [ "$FORCE" = 1 ] || [ ! -f "$HERE/obj/$f.o" ] || [ "$(cat "$HERE/obj/$f.key" 2>/dev/null)" != "$key" ]; then
    rm -f "$HERE/obj/$f.key"
    ( $HIPCC $FLAGS $EXTRA -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" && echo "$key" > "$HERE/obj/$f.key" ) &
    pids+=($!)
    names+=("$f")
  fi
done
fail=0
for i in "${!pids[@]}"; do wait "${pids[$i]}" || { echo "build.sh: compiling ${names[$i]}.hip failed" >&2; fail=1; }; done
[ "$fail" = 0 ] || exit 1
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/obj/*.o
echo "built $OUT (${#pids[@]} object(s) recompiled)"
