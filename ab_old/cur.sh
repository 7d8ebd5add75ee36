#!/bin/bash
set -e
python -m pytest tests/test_gpu_backward.py -q -x > gpurun_out/t1.log 2>&1
python tools_bench_bwd.py --iters 5 > gpurun_out/bwd_cur.json 2> gpurun_out/bwd_cur.err
