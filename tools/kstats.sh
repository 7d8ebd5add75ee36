#!/bin/bash
# Register / spill / scratch numbers of every kernel in one csrc file (device-only asm, gfx950).  tools/kstats.sh lifting_f16
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../imagecompressionlearnedliftingandlearnedtreebasedmodels_amd/csrc/$1.hip"
OUT="${TMPDIR:-/tmp}/kstats_$1.s"
EXTRA=""
[ "$1" = conv_f16x3 ] && EXTRA="-mllvm -pragma-unroll-threshold=131072"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $EXTRA -S --cuda-device-only "$SRC" -o "$OUT" -Wno-unused 2>&1 | grep -E "error" 
grep -E "^\s+\.(sgpr_count|vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|name):" "$OUT" | sed 's/^ *//' | paste -sd' ' | sed 's/\.name:/\n.name:/g' | awk 'NF{print $2, $3,$4,$5,$6,$7,$8,$9,$10,$11,$12,$13,$14}'
echo "asm: $OUT"
