#!/bin/bash
# Kernel-trace/stats pass of the training leg (run on the GPU box: bash tools/profile_train.sh r03): 1 eval step + 3 training steps.
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_train -o st -- python3 $R/bench.py --steps 1 --warmup 1 --train-steps 3 --no-cpu-baseline --no-hbm-kernels > $R/gpurun_out/${TAG}_train.log 2>&1
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob("gpurun_out/%s_train/**/st_kernel_stats.csv" % tag, recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/%s_train_kernel_stats_top.txt" % tag, "w") as o:
    o.write("total kernel time %.1f ms over the run (1+1 eval steps, 2 warm-up + 3 timed training steps)\n" % (tot / 1e6))
    for r in rows[:40]:
        o.write("%8.2f ms %6s calls %5.1f%%  %s\n" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:150]))
print(open("gpurun_out/%s_train_kernel_stats_top.txt" % tag).read())
PY
