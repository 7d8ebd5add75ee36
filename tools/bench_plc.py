#!/usr/bin/env python
"""Micro-benchmark of the tree-context conv 243 -> 243 3x3: fp32 MFMA engine vs split-fp16 (f16x3), level-0 shape of
BASELINE configs[2] (3 planes x 8 images x 256 x 256).  Prints one JSON line.   python tools/bench_plc.py [--iters 5]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    dev = "cuda:0"
    P, B, C, S = 3, a.batch, 243, a.size
    x = torch.randn(P, B, C, S, S, device=dev).clamp_(min=-0.5)
    w = (torch.rand(P, C, C, 3, 3, device=dev) - 0.5) * 0.04
    b = torch.rand(P, C, device=dev) - 0.5
    flop = 2.0 * C * C * 9 * P * B * S * S
    pk32 = ops.conv_pack(w, 3)
    pk16 = ops.conv_f16x3_pack(w)
    out = {"shape": [P, B, C, S, S], "flop": flop}
    t = timeit(lambda: ops.conv2d(x, w, b, 3, packed=pk32), a.iters)
    out["f32"] = {"ms": t * 1e3, "TFLOP/s": flop / t / 1e12}
    slots = ops.absmax_slots(x)
    t = timeit(lambda: ops.conv3x3_f16x3(x, pk16, b, C, slots=slots), a.iters)
    out["f16x3"] = {"ms": t * 1e3, "TFLOP/s (algorithmic fp32-equivalent)": flop / t / 1e12, "issued TFLOP/s": 3 * flop * (256 / 243) ** 2 / t / 1e12}
    t = timeit(lambda: ops.absmax_slots(x), a.iters)
    out["absmax_slots"] = {"ms": t * 1e3, "GB/s": x.numel() * 4 / t / 1e9}
    # the PAIR (first tree conv 3 -> 243 on the 2x-upsampled parent + the dense conv): two launches vs one fused launch
    parent = torch.randn(P, B, 3, S // 2, S // 2, device=dev).round_()
    w1 = (torch.rand(P, C, 3, 3, 3, device=dev) - 0.5) * 0.4
    b1 = torch.rand(P, C, device=dev) - 0.5
    pk1 = ops.conv_pack(w1, 3)
    pk1f = ops.plc_fused_pack1(w1, b1)

    def two():
        sl = torch.empty(P, 64, device=dev)
        t1 = ops.conv2d(parent, w1, b1, 3, act=ops.ACT_LRELU, upsample2=True, packed=pk1, absmax=sl)
        return ops.conv3x3_f16x3(t1, pk16, b, C, slots=sl)
    t = timeit(two, a.iters)
    out["pair_two_launches"] = {"ms": t * 1e3}
    t = timeit(lambda: ops.plc_fused(parent, pk1f, pk16, b, C, C), a.iters)
    out["pair_fused"] = {"ms": t * 1e3, "TFLOP/s (algorithmic fp32-equivalent, second conv only)": flop / t / 1e12}
    out["pair_max_abs_diff"] = float((two() - ops.plc_fused(parent, pk1f, pk16, b, C, C)).abs().max())
    # backward-weights of the same layer: fp32 MFMA kernel vs split-fp16
    dyg = torch.randn(P, B, C, S, S, device=dev) * 1e-3
    t = timeit(lambda: ops.conv2d_wgrad(x, dyg, (P, C, C, 3, 3), 3), a.iters)
    out["wgrad_f32"] = {"ms": t * 1e3, "TFLOP/s": flop / t / 1e12}
    t = timeit(lambda: ops.conv3x3_wgrad_f16x3(x, dyg, (P, C, C, 3, 3)), a.iters)
    out["wgrad_f16x3"] = {"ms": t * 1e3, "TFLOP/s (algorithmic fp32-equivalent)": flop / t / 1e12}
    y32 = ops.conv2d(x, w, b, 3, packed=pk32)
    y16 = ops.conv3x3_f16x3(x, pk16, b, C, slots=slots)
    out["max_abs_diff_f16x3_vs_f32"] = float((y32 - y16).abs().max())
    out["out_scale"] = float(y32.abs().max())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
