#!/bin/bash
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_backward.py -q -x > gpurun_out/t1.log 2>&1
python bench.py --steps 5 --warmup 2 --train-steps 3 --no-cpu-baseline > gpurun_out/b_train.log 2>&1
