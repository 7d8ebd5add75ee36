#!/bin/bash
set -e
python -m pytest tests -q -m gpu -x -k "cdf97 or CDF97" > gpurun_out/t1.log 2>&1
python tools_bench_kernels.py --batch 8 --iters 20 > gpurun_out/k8.log 2>&1
python tools_bench_kernels.py --batch 96 --iters 5 > gpurun_out/k96.log 2>&1
