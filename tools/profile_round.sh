#!/bin/bash
# Profile set of a round (run on the GPU box: bash tools/profile_round.sh r03).  One kernel-trace/stats pass and SEPARATE --pmc
# passes (FETCH_SIZE; WRITE_SIZE; SQ busy / MFMA busy) over the same bench.py command, as the MI355X guide prescribes, plus a
# second stats pass with twice the steps (does a per-launch count scale with the steps, or is it set-up?).
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
ARGS="--steps 3 --warmup 1 --train-steps 0 --no-cpu-baseline --no-hbm-kernels"
ARGS1="--steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline --no-hbm-kernels"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o st -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats8 -o st -- python3 $R/bench.py --steps 7 --warmup 1 --train-steps 0 --no-cpu-baseline --no-hbm-kernels > $R/gpurun_out/${TAG}_stats8.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -o f -- python3 $R/bench.py $ARGS1 > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -o w -- python3 $R/bench.py $ARGS1 > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/${TAG}_sq -o s -- python3 $R/bench.py $ARGS1 > $R/gpurun_out/${TAG}_sq.log 2>&1
cd $R
python3 tools/profile_round_summary.py $TAG
