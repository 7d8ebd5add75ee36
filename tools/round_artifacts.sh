#!/bin/bash
# Everything a round commits under profiles/ besides the rocprofv3 passes of tools/profile_round.sh (run on the GPU box:
# bash tools/round_artifacts.sh r03).  Results land in gpurun_out/<tag>_*; copy the summaries to profiles/ afterwards.
TAG=${1:-r03}
O=gpurun_out
python bench.py > $O/${TAG}_d_bench_cfg3.json 2> $O/${TAG}_d_bench_cfg3.err
for c in 0 1 3 4; do python bench.py --config $c --no-cpu-baseline --no-hbm-kernels > $O/${TAG}_g_bench_cfg_idx$c.json 2> $O/${TAG}_g_$c.err; done
for c in 1 4; do python bench.py --config $c --precision f16x3 --train-steps 0 --no-cpu-baseline --no-hbm-kernels > $O/${TAG}_g_bench_cfg_idx${c}_f16x3.json 2>> $O/${TAG}_g_$c.err; done
python tools/lift_stamps.py > $O/${TAG}_h_lift_stamps.json 2> $O/${TAG}_h.err
python tools/bench_lift.py 2> $O/${TAG}_j.err | tail -1 > $O/${TAG}_j_lift_microbench.json
python tools/time_coding.py 2>&1 | grep compress > $O/${TAG}_p_coding_time.txt
python tools/dbg/wg16.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_q_wgrad16_microbench.json
python -m pytest tests/test_gpu_fullsize_oracle.py tests/test_gpu_precision.py tests/test_gpu_train.py -q -m gpu -s 2>&1 | grep -E "^\[|passed|failed" > $O/${TAG}_e_parity.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/${TAG}_train_stats -o st -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --train-steps 4 --no-cpu-baseline --no-hbm-kernels > $GRAFT_REPO_ROOT/$O/${TAG}_train_stats.log 2>&1
echo done
