#!/bin/bash
# A/B over library variants (box copy only)
set -e
P=imagecompressionlearnedliftingandlearnedtreebasedmodels_amd
cp $P/liblldwt.so /tmp/keep.so
for f in ab_old/*.so; do
  n=$(basename $f .so)
  cp $f $P/liblldwt.so
  python tools_bench_bwd.py --iters 5 > gpurun_out/bwd_$n.json 2> gpurun_out/bwd_$n.err
done
cp /tmp/keep.so $P/liblldwt.so
