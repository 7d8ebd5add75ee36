"""Debug: backward of one lifting step, fused split-fp16 BWD mode against the fp32-MFMA launches (same inputs)."""
import sys
import torch
from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops

torch.manual_seed(0)
dev = torch.device("cuda:0")
P, B, C, K = 3, 2, 16, 5
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40, 72)
Z, n = P * B, P * B * h * w
W = {"w1": torch.randn(P, C, 1, K, K) * 0.2, "b1": torch.randn(P, C) * 0.1, "w2": torch.randn(P, C, C, K, K) * 0.05,
     "b2": torch.randn(P, C) * 0.1, "w3": torch.randn(P, C, C, K, K) * 0.05, "b3": torch.randn(P, C) * 0.1,
     "w4": torch.randn(P, 1, C, K, K) * 0.05, "b4": torch.randn(P, 1) * 0.1}
W = {k: v.to(dev).contiguous() for k, v in W.items()}
keys = ("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4")
packed = ops.pack_pblock(*[W[k] for k in keys])
bpack = ops.pack_pblock_bwd(W["w1"], W["w2"], W["w3"], W["w4"])
taps = torch.tensor([0.1, 0.8, 0.1], device=dev).repeat(P, 1).contiguous()
tid = torch.tensor([0.0, 1.0, 0.0], device=dev).repeat(P, 1).contiguous()
saved = torch.empty(n * (2 + 3 * C), device=dev)
saved[:2 * n] = torch.randn(2 * n, device=dev)
saved[2 * n:2 * n + 2 * n * C] = torch.tanh(torch.randn(2 * n * C, device=dev))
saved[2 * n + 2 * n * C:] = torch.randn(n * C, device=dev)
gout = torch.randn(Z, h, w, device=dev)
import ctypes
for vert in (1, 0):
    res = []
    for fused in (0, 1):
        gdin, gsrc = torch.zeros(Z, h, w, device=dev), torch.zeros(Z, h, w, device=dev)
        dW = [torch.zeros_like(W[k]) for k in keys]
        dtaps = torch.zeros_like(taps)
        v = lambda t: ops.View(ctypes.c_void_p(t.data_ptr()), h * w, w, 1)
        ops.lift_step_bwd(v(gout), v(gdin), v(gsrc), saved, P, B, h, w, taps, dtaps, ctypes.c_void_p(packed.data_ptr()),
                          packed.shape[1], dW, C, K, 0.5, -1.0, vert, False,
                          packed_bwd=ctypes.c_void_p(bpack.data_ptr()) if fused else None, taps_id=tid if fused else None)
        torch.cuda.synchronize()
        ws = ops.workspace(0, dev).view(torch.float32)
        parts = {"g": ws[:n], "dsk": ws[n:2 * n], "dt3": ws[2 * n:2 * n + n * C], "dpre2": ws[2 * n + n * C:2 * n + 2 * n * C],
                 "dr": ws[2 * n + 2 * n * C:2 * n + 3 * n * C]}
        res.append(({k: t.clone() for k, t in parts.items()}, gdin, gsrc, dW, dtaps))
    a, b = res
    for k in a[0]:
        d = (a[0][k] - b[0][k]).abs().max().item()
        print("vert", vert, k, "max|ref|", a[0][k].abs().max().item(), "maxdiff", d)
    print("gsrc", (a[2] - b[2]).abs().max().item(), "dtaps", (a[4] - b[4]).abs().max().item())
    for k, x, y in zip(keys, a[3], b[3]):
        print(" ", k, "rel", ((x - y).norm() / (x.norm() + 1e-30)).item())
