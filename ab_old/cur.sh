#!/bin/bash
set -e
python -m pytest tests/test_gpu_lifting.py tests/test_gpu_backward.py -q -x > gpurun_out/t1.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fwd -o fwd --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --train-steps 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_fwd.log 2>&1
cd $GRAFT_REPO_ROOT && python bench.py --steps 5 --warmup 2 --train-steps 3 --no-cpu-baseline > gpurun_out/b_train.log 2>&1
