"""Sweep a process-wide switch of the HBM-bound kernels: python tools/dbg/rate_sweep.py LLDWT_CDF_TILE 32 1632 default (one child process per value;
several switches at once: A,B 1,2 3,4;
the secondary roofline block of bench.py in each)."""
import json
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import torch
    import bench
    dev = torch.device("cuda:0")
    for B in (8, 96):
        for r in bench.hbm_kernels(dev, B, 512):
            print(json.dumps({"B": B, "kernel": r["kernel"], "frac": round(r["frac"], 4), "us": round(r["ms"] * 1e3, 2)}), flush=True)
else:
    var, vals = sys.argv[1], sys.argv[2:]
    for v in vals:
        print("==", var, v, flush=True)
        env = dict(os.environ)
        if v != "default":
            for name, val in zip(var.split(","), v.split(",")):     # several switches at once: A,B 1,2 3,4
                env[name] = val
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=True)
