#!/bin/bash
set -e
python -m pytest tests/test_gpu_backward.py -q -x -k "wgrad or lifting" > gpurun_out/t1.log 2>&1
python tools_bench_bwd.py --iters 5 --only lift > gpurun_out/bwd_cur.json 2> gpurun_out/bwd_cur.err
python bench.py --steps 3 --warmup 1 --train-steps 3 --no-cpu-baseline > gpurun_out/b_train.log 2>&1
