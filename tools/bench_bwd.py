#!/usr/bin/env python
"""Micro-benchmark of the backward kernels on the shapes of BASELINE configs[2] (P=3 planes, B=8, 512x512, L=4):
weight-gradient GEMMs and backward-data convs of the lifting P-blocks, the tree context conv and the cgp 1x1 stack.
Prints one JSON object: per case ms and useful TFLOP/s (2*MACs of the layer, not the padded MFMA work).

    python tools/bench_bwd.py [--iters 10] [--only lift]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    dev = torch.device("cuda:0")
    P, B = 3, 8
    out = {}
    # (name, cin, cout, K, groups, h, w)
    cases = [
        ("lift_16x16_k5_L0row", 16, 16, 5, 1, 256, 512),
        ("lift_16x16_k5_L2col", 16, 16, 5, 1, 64, 64),
        ("lift_1x16_k5_L0row", 1, 16, 5, 1, 256, 512),
        ("lift_16x1_k5_L0row", 16, 1, 5, 1, 256, 512),
        ("plc_243x243_k3_L0", 243, 243, 3, 1, 256, 256),
        ("plc_3x243_k3_L0", 3, 243, 3, 1, 256, 256),
        ("cgp_486x486_k1_g3_L0", 486, 486, 1, 3, 256, 256),
        ("cgp_486x162_k1_g3_L0", 486, 162, 1, 3, 256, 256),
    ]
    for name, cin, cout, K, g, h, w in cases:
        if a.only and a.only not in name:
            continue
        x = torch.randn(P, B, cin, h, w, device=dev)
        dy = torch.randn(P, B, cout, h, w, device=dev)
        wt = torch.randn(P, cout, cin // g, K, K, device=dev) * 0.05
        flop = 2.0 * P * B * h * w * cout * (cin // g) * K * K
        dw = torch.zeros_like(wt)
        db = torch.zeros(P, cout, device=dev)
        t = timeit(lambda: ops.conv2d_wgrad(x, dy, tuple(wt.shape), K, groups=g, dw=dw, db=db), a.iters)
        out[name + ":wgrad"] = {"ms": t * 1e3, "TFLOP/s": flop / t / 1e12}
        pk = ops.conv_pack(wt, K, groups=g, transposed=True)
        t = timeit(lambda: ops.conv2d(dy, wt, None, K, groups=g, transposed=True, packed=pk), a.iters)
        out[name + ":bwd_data"] = {"ms": t * 1e3, "TFLOP/s": flop / t / 1e12}
        del x, dy
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
