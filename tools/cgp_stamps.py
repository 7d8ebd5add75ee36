#!/usr/bin/env python
"""Diagnostic: in-kernel clock stamps of the cgp register-chain kernel (k_cgp16) at the level-0 shape of BASELINE
configs[2] (3 planes x 8 images x 3 subbands of 256 x 256).   python tools/cgp_stamps.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    dev = "cuda:0"
    P, B, G, S = 3, 8, 3, 256
    torch.manual_seed(0)
    c = [93, 162, 54, 18, 2]
    ws = [(torch.randn(P, G * c[i + 1], c[i], 1, 1, device=dev) / c[i] ** 0.5) for i in range(4)]
    bs = [torch.randn(P, G * c[i + 1], device=dev) * 0.1 for i in range(4)]
    packed16 = ops.cgp16_pack(ws, bs, G)
    plc = torch.randn(P, B, G * 81, S, S, device=dev)
    xq = torch.randn(P, B, G, S, S, device=dev).round_()
    tap_bits = 0b0000000000000_11_11111_11111          # the 12 causal taps of the 5x5 type-A mask

    def run():
        return ops.cgp16_params(plc, xq, packed16, 5, tap_bits)
    for _ in range(30):
        run()
    torch.cuda.synchronize()
    cols = S * S // 32
    st = torch.zeros(P * B, G, cols, 8, dtype=torch.int64, device=dev)
    ops.set_diagnostics(2, st)
    run()
    torch.cuda.synchronize()
    ops.set_diagnostics(2, None)
    s = st.cpu().numpy().astype(np.int64)
    d = np.diff(s[..., :6], axis=-1)
    names = ["inputs (48 loads per lane per block) + |max|", "layers 0 + 1 (60 weight steps, 360 MFMAs)", "layer 2", "layer 3", "store"]
    res = {"mean_cycles_per_wave": {n: float(d[..., i].mean()) for i, n in enumerate(names)}}
    tot = s[..., 5] - s[..., 0]
    real = (s[..., 7] - s[..., 6]).astype(np.float64)
    ok = real > 0
    res["total_cycles_mean"] = float(tot.mean())
    res["ideal_mfma_cycles"] = 66 * 2 * 3 * 32
    res["in_kernel_clock_GHz"] = float(np.median(tot[ok] / real[ok]) * 0.1)
    t0, t1 = s[..., 6].min(), s[..., 7].max()
    res["launch_span_us"] = float((t1 - t0) / 100.0)
    res["waves"] = int(s[..., 0].size)
    res["mean_wave_duration_us"] = float(real[ok].mean() / 100.0)
    res["resident_waves_estimate"] = float(real[ok].sum() / (t1 - t0))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
