#!/usr/bin/env python
"""Micro-benchmark of ONE lifting step at the level-0 row-pass shape of BASELINE configs[2] (24 planes*images of
256 x 512): fused split-fp16 kernel vs the three fp32-MFMA launches; the debug mask (ops.set_diagnostics) skips phases for
attribution (1 = conv1, 2 = conv2, 16 = sequential conv3 / conv4, 32 = no vertical reuse)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import _lib, ops
    lib = _lib.load()
    dev = "cuda:0"
    P, B, h, w = 3, 8, 256, 512
    torch.manual_seed(0)
    x = torch.rand(P * B, 2 * h, w, device=dev) - 0.5
    ws = [(torch.randn(P, *s, device=dev) * sc) for s, sc in (((16, 1, 5, 5), 0.2), ((16,), 0.1), ((16, 16, 5, 5), 0.05), ((16,), 0.1),
                                                             ((16, 16, 5, 5), 0.05), ((16,), 0.1), ((1, 16, 5, 5), 0.05), ((1,), 0.1))]
    packed = ops.pack_pblock(*ws)
    taps = torch.tensor([0.0, -1.586, -1.586], device=dev).repeat(P, 1).contiguous()
    src = ops.view_of(x, P * B, h, w, offset=0, sz=2 * h * w, sy=2 * w, sx=1)
    din = ops.view_of(x, P * B, h, w, offset=w, sz=2 * h * w, sy=2 * w, sx=1)
    out = torch.empty(P * B, h, w, device=dev)
    dout = ops.view_of(out, P * B, h, w)

    def run():
        ops.lift_step(src, din, dout, P * B, B, h, w, taps, packed, 16, 5, True, 1.0, 0.1)

    def timeit(n=10):
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    res = {}
    lib.lldwt_set_lift_mode(0)
    res["f32_three_launches_us"] = timeit()
    ref = out.clone()
    lib.lldwt_set_lift_mode(1)
    for dbg in (0, 32, 16, 1, 2, 3, 1 + 32, 2 + 32, 3 + 32):
        ops.set_diagnostics(0, None, dbg)
        res["fused_dbg%d_us" % dbg] = timeit()
    ops.set_diagnostics(0, None, 0)
    run()
    res["max_abs_diff_vs_f32"] = float((out - ref).abs().max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
