#!/usr/bin/env python
"""Diagnostic: in-kernel clock stamps (s_memtime) of the fused lifting step at the level-0 row-pass shape of BASELINE
configs[2]: per-phase cycles of every wave, barrier waits, in-kernel clock.  The kernel writes stamps only when
a stamp buffer is registered through lldwt_set_diagnostics (this tool); a normal run executes none.   python tools/lift_stamps.py"""
import json
import os
import sys

import numpy as np
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
    lib.lldwt_set_lift_mode(1)
    nwg = (w // 32) * (h // 16) * P * B
    st = torch.zeros(nwg, 8, 16, dtype=torch.int64, device=dev)

    def run():
        ops.lift_step(src, din, dout, P * B, B, h, w, taps, packed, 16, 5, True, 1.0, 0.1)
    for _ in range(200):                       # warm the clocks under load
        run()
    torch.cuda.synchronize()
    abl = None
    dbg = int(os.environ.get("LLDWT_LF_DBG", "0"))
    for _ in range(20):
        run()
    ops.set_diagnostics(0, st, dbg)
    run()
    torch.cuda.synchronize()
    ops.set_diagnostics(0, None, 0)
    s = st.cpu().numpy().astype(np.int64)
    gx, gy = w // 32, h // 16
    s = s.reshape(P * B, gy, gx, 8, 16)
    interior = np.zeros((gy, gx), bool)
    interior[1:-1, 1:-1] = True                # y0 >= 2 and y0 + 18 <= h etc: every tile but the frame
    res = {"ablations_interior_mean_cycles": abl}
    si = s[:, interior]                        # (Z, n_int, 8, 16)
    names = ["P0 load+filter", "P0 barrier", "P1 conv1+tanh", "P1 barrier", "P2 conv2+tanh", "P2 barrier",
             "PC composite", "PC barrier", "PC finish+store"]
    d = np.diff(si[..., :10], axis=-1)
    res["interior_mean_cycles_per_wave"] = {n: float(d[..., i].mean()) for i, n in enumerate(names)}
    res["interior_max_over_waves_mean"] = {n: float(d[..., i].max(axis=-1).mean()) for i, n in enumerate(names)}
    tot = si[..., 9] - si[..., 0]
    res["interior_total_cycles_mean"] = float(tot.mean())
    # vertical reuse: with a run length of RL tiles (LLDWT_STAMPS_RL, default 8 at this shape) the tiles at ty % RL == 0 start a
    # run (all 28 / 24 rows of t1 / t2), the others continue one (16 new rows each)
    rl = int(os.environ.get("LLDWT_STAMPS_RL", "8"))
    ty = np.arange(gy)[:, None] * np.ones((1, gx), int)
    for name, sel in (("first_of_run", interior & (ty % rl == 0)), ("continuing", interior & (ty % rl != 0))):
        if sel.any():
            sx_ = s[:, sel]
            dd = np.diff(sx_[..., :10], axis=-1)
            res["interior_%s_mean_cycles_per_wave" % name] = {n: float(dd[..., i].mean()) for i, n in enumerate(names)}
            res["interior_%s_total" % name] = float((sx_[..., 9] - sx_[..., 0]).mean())
    real = (si[..., 15] - si[..., 14]).astype(np.float64)      # 100 MHz ticks
    ok = real > 0
    res["in_kernel_clock_GHz"] = float(np.median(tot[ok] / real[ok]) * 0.1)
    res["wg_duration_us_median"] = float(np.median(real[ok]) / 100.0)
    sb = s[:, ~interior]
    if os.environ.get("LLDWT_LF_DBG", "0") == "16":          # sequential path for every tile
        res["border_total_cycles_mean"] = float((sb[..., 12] - sb[..., 0]).mean())
    else:                                                     # composed path + strip correction: same stamps as interior tiles
        db = np.diff(sb[..., :10], axis=-1)
        res["border_mean_cycles_per_wave"] = {n: float(db[..., i].mean()) for i, n in enumerate(names)}
        res["border_total_cycles_mean"] = float((sb[..., 9] - sb[..., 0]).mean())
        edge = np.zeros((gy, gx), bool)
        edge[0, 1:-1] = True
        se = s[:, edge]
        res["top_edge_total_cycles_mean"] = float((se[..., 9] - se[..., 0]).mean())
        res["top_edge_finish"] = float((se[..., 9] - se[..., 8]).mean())
        edge[:] = False
        edge[1:-1, 0] = True
        se = s[:, edge]
        res["left_edge_total_cycles_mean"] = float((se[..., 9] - se[..., 0]).mean())
        res["left_edge_finish"] = float((se[..., 9] - se[..., 8]).mean())
        res["left_edge_PC+strips"] = float((se[..., 7] - se[..., 6]).mean())
    # the span of the launch in real time and the number of workgroups resident at once
    t0, t1 = s[..., 14].min(), s[..., 15].max()
    res["launch_span_us"] = float((t1 - t0) / 100.0)
    res["sum_wg_duration_over_span_per_cu"] = float(((s[..., 0, 15] - s[..., 0, 14]).sum() / 100.0) / ((t1 - t0) / 100.0) / 256)
    # per-CU timeline: gap between a workgroup's end and the next one's start on the same CU
    hw = s[..., 0, 13].reshape(-1)
    xcc, cu, se = (hw >> 32) & 0xF, (hw >> 8) & 0xF, (hw >> 13) & 0x7
    key = (xcc * 8 + se) * 16 + cu
    t_start, t_end = s[..., 0, 14].reshape(-1), s[..., :, 15].max(axis=-1).reshape(-1)
    gaps, per_cu = [], []
    for k in np.unique(key):
        m = key == k
        o = np.argsort(t_start[m])
        ts, te = t_start[m][o], t_end[m][o]
        gaps.extend(((ts[1:] - te[:-1]) / 100.0).tolist())
        per_cu.append(int(m.sum()))
    res["distinct_cus"] = int(len(np.unique(key)))
    res["wgs_per_cu_min_max"] = [min(per_cu), max(per_cu)]
    res["gap_us_between_wgs_on_a_cu"] = {"median": float(np.median(gaps)), "mean": float(np.mean(gaps)),
                                         "p90": float(np.percentile(gaps, 90))}
    res["first_start_spread_us"] = float((np.sort(t_start)[255] - t_start.min()) / 100.0)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
