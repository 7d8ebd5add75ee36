"""profiles/<tag>_k_kernel_stats_train_cfg3.csv from the kernel-trace/stats pass of the training leg that tools/round_artifacts.sh
leaves under gpurun_out/<tag>_train_stats (bench.py --steps 1 --warmup 1 --train-steps 4: 2 warm-up + 4 timed training steps):
    python tools/train_stats_summary.py r03"""
import csv
import glob
import os
import sys

from profile_round_summary import R, OUT, short

TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
f = glob.glob(os.path.join(R, "gpurun_out", TAG + "_train_stats", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(os.path.join(OUT, TAG + "_k_kernel_stats_train_cfg3.csv"), "w") as o:
    w = csv.writer(o)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage",
                "ms_per_training_step(6 steps: 2 warm-up + 4 timed; eval kernels of the 2 eval steps included in the totals)"])
    for r in rows[:45]:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                    "%.3f" % (float(r["TotalDurationNs"]) / 6e6)])
print("total kernel time per training step: %.1f ms" % (sum(float(r["TotalDurationNs"]) for r in rows) / 6e6))
