#!/usr/bin/env python
"""Diagnostic: in-kernel clock stamps (s_memtime) of the fused tree-context pair (k_conv3_f16x3<2>) at the level-0 shape of
BASELINE configs[2] (3 planes x 8 images, parent 128 x 128 -> 243 channels at 256 x 256).  The kernel writes stamps only
when a stamp buffer is registered through lldwt_set_diagnostics (this tool).   python tools/plc_stamps.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    dev = "cuda:0"
    P, B, C, S = 3, 8, 243, 256
    torch.manual_seed(0)
    parent = torch.randn(P, B, 3, S // 2, S // 2, device=dev).round_()
    w1 = (torch.rand(P, C, 3, 3, 3, device=dev) - 0.5) * 0.4
    b1 = torch.rand(P, C, device=dev) - 0.5
    w2 = (torch.rand(P, C, C, 3, 3, device=dev) - 0.5) * 0.04
    b2 = torch.rand(P, C, device=dev) - 0.5
    pk1 = ops.plc_fused_pack1(w1, b1)
    pk2 = ops.conv_f16x3_pack(w2)

    def run():
        return ops.plc_fused(parent, pk1, pk2, b2, C, C)
    for _ in range(100):
        run()
    torch.cuda.synchronize()
    nwg = (S // 32) * (S // 8) * 2 * P * B
    st = torch.zeros(nwg, 4, 16, dtype=torch.int64, device=dev)
    ops.set_diagnostics(1, st)
    run()
    torch.cuda.synchronize()
    ops.set_diagnostics(1, None)
    s = st.cpu().numpy().astype(np.int64)
    res = {"workgroups": nwg}
    d = np.diff(s[..., :12], axis=-1)
    names = ["prologue (gather, scales, chunk 0 staging, ring fill)"] + ["chunk %d" % i for i in range(8)] + ["epilogue stores"]
    cols = [0] + list(range(1, 9)) + [10 - 0]
    res["mean_cycles_per_wave"] = {"prologue": float(d[..., 0].mean()), "chunks": [float(d[..., 1 + i].mean()) for i in range(8)],
                                   "last chunk -> loop end": float(d[..., 9].mean()), "epilogue": float(d[..., 10].mean())}
    res["ideal_cycles_per_chunk"] = 36 * 12 * 32 + 18 * 32
    tot = s[..., 11] - s[..., 0]
    real = (s[..., 15] - s[..., 14]).astype(np.float64)
    ok = real > 0
    res["total_cycles_mean"] = float(tot.mean())
    res["in_kernel_clock_GHz"] = float(np.median(tot[ok] / real[ok]) * 0.1)
    res["wg_duration_us_median"] = float(np.median(real[ok]) / 100.0)
    t0, t1 = s[..., 14].min(), s[..., 15].max()
    res["launch_span_us"] = float((t1 - t0) / 100.0)
    res["sum_wg_duration_over_span_per_cu"] = float(((s[:, 0, 15] - s[:, 0, 14]).sum() / 100.0) / ((t1 - t0) / 100.0) / 256)
    res["wave_skew_at_loop_end_cycles"] = float((s[..., 10].max(axis=-1) - s[..., 10].min(axis=-1)).mean())
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
