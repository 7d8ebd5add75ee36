"""Debug: the training cgp stack alone at the level-0 shape of configs[2] (3 x 8 x 3 x 256 x 256): forward (register chain + stores)
and backward (gauss_rate_bwd, backward-data chain, four weight gradients, taps transpose).   python tools/dbg/cgp_train.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import autograd as ag
dev = "cuda:0"
torch.manual_seed(0)
P, B, G, h, w, K = 3, 8, 3, 256, 256, 5
c = [93, 162, 54, 18, 2]
tap_mask = (1 << 12) - 1
plc = torch.randn(P, B, G * 81, h, w, device=dev).requires_grad_(True)
xq = torch.round(torch.randn(P, B, G, h, w, device=dev) * 3).requires_grad_(True)
x = (torch.randn(P, B, G, h, w, device=dev) * 2).requires_grad_(True)
noise = torch.rand(P, B, G, h, w, device=dev) - 0.5
ws = [(torch.randn(P, G * c[l + 1], c[l], 1, 1, device=dev) * (1.5 / c[l] ** 0.5)).requires_grad_(True) for l in range(4)]
bs = [(torch.randn(P, G * c[l + 1], device=dev) * 0.1).requires_grad_(True) for l in range(4)]
wb = [t for pair in zip(ws, bs) for t in pair]
gb = torch.rand(P, B, G, h, w, device=dev)
def run():
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    bits = ag.CgpRateCtxFn.apply(plc, xq, x, noise, G, K, tap_mask, *wb)
    e[1].record()
    bits.backward(gb)
    e[2].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
for _ in range(2):
    run()
f = b = 0.0
for _ in range(5):
    a_, b_ = run()
    f += a_ / 5
    b += b_ / 5
print("cgp stack at 3x8x3x256x256: forward %.3f ms, backward %.3f ms" % (f, b))
