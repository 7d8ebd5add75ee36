"""Debug: the roofline_hbm rows of bench.py alone."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0)
for B in (8, 96):
    for r in bench.hbm_kernels(dev, B, 512):
        print("%-45s %8.1f GB/s  frac %.3f  %.2f us (eager %.2f us)" % (r["kernel"] + " @%d" % B, r["achieved"], r["frac"], r["ms"] * 1e3, r["ms_eager"] * 1e3))
