#!/bin/bash
# Two separate rocprofv3 --pmc passes (SQ wave/wait/LDS counters, then instruction mix / MFMA busy) over bench.py.
# Run on the GPU box: bash tools/pmc_bench.sh ; summarise with python tools/pmc_show.py <kernel-name-substring>...
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/pmc_w1 -o w1 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline > $R/gpurun_out/pmc_w1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_w2 -o w2 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline > $R/gpurun_out/pmc_w2.log 2>&1
