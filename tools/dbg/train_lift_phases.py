"""Debug: training forward / backward of the 4-level lifting transform (configs[2] shape) with phases of the fused kernel masked
(lldwt_set_diagnostics flags: 1 = no conv1, 2 = no conv2, 4 = no conv3, 8 = no conv4).  Measured at the end of round 3 (one
transform): forward 7.85 ms = conv1 1.26 + conv2 2.36 + conv3 1.64 + conv4 0.69 + 2.35 with all four masked (skip patches, staging,
barriers, the small levels' launches; the stores of the saved tensors are 0.55 of it); backward 20.4 ms, 14.5 with the fused
backward kernel's four phases masked (weight gradients ~11, the kernel's remainder 2.4, skip-filter transpose 1.2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import autograd as ag, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
P, B, H, W, L, nb = 3, 8, 512, 512, 4, 2
meta = dict(levels=L, C=16, K=5, rw=0.1, linear=False, different=False)
taps = torch.tensor([[0.1, 0.8, 0.1]], device=dev).repeat(4 * P, 1).view(4, P, 3).contiguous().requires_grad_(True)
shapes = [(16, 1, 5, 5), (16,), (16, 16, 5, 5), (16,), (16, 16, 5, 5), (16,), (1, 16, 5, 5), (1,)]
Wt = [(torch.randn(nb, 2, P, *s, device=dev) * 0.05).requires_grad_(True) for s in shapes]
x = (torch.rand(P, B, 1, H, W, device=dev) - 0.5).requires_grad_(True)

def run(flags, n=3):
    ops.set_diagnostics(0, None, flags)
    tf = tb = 0.0
    for i in range(n + 1):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        outs = ag.LiftingFn.apply(x, taps, meta, None, None, *Wt)
        e[1].record()
        torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
        e[2].record()
        torch.cuda.synchronize()
        if i:
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    return tf / n, tb / n
try:
    for fl in (0, 1, 2, 4, 8, 1 | 2 | 4 | 8):
        f, b = run(fl)
        print("flags %3d: forward %.2f ms, backward %.2f ms" % (fl, f, b))
finally:
    ops.set_diagnostics(0, None, 0)
