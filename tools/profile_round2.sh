#!/bin/bash
# Round-2 profile set (run on the GPU box: bash tools/profile_round2.sh).  One kernel-trace/stats pass and separate --pmc
# passes (FETCH_SIZE; WRITE_SIZE; SQ busy / MFMA busy) over the same bench.py command, as the MI355X guide prescribes.
set -e
R=$GRAFT_REPO_ROOT
ARGS="--steps 3 --warmup 1 --train-steps 0 --no-cpu-baseline --no-hbm-kernels"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_stats -o st -- python3 $R/bench.py $ARGS > $R/gpurun_out/r02_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline --no-hbm-kernels > $R/gpurun_out/r02_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02_write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline --no-hbm-kernels > $R/gpurun_out/r02_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r02_sq -o s -- python3 $R/bench.py --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline --no-hbm-kernels > $R/gpurun_out/r02_sq.log 2>&1
cd $R
python3 tools/profile_round2_summary.py
