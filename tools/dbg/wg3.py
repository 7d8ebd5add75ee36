"""Debug: the split-fp16 weight gradient of the 243 -> 243 tree conv alone (level-0 shape of configs[2]) -- for rocprofv3 --pmc
runs and timing of the kernel without the |max| passes.   python tools/dbg/wg3.py [batch] [size] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
it = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda:0"
P, C = 3, 243
x = torch.randn(P, B, C, S, S, device=dev).clamp_(min=-0.5)
dy = torch.randn(P, B, C, S, S, device=dev) * 1e-3
for _ in range(2):
    ops.conv3x3_wgrad_f16x3(x, dy, (P, C, C, 3, 3))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(it):
    ops.conv3x3_wgrad_f16x3(x, dy, (P, C, C, 3, 3))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / it
fl = 2.0 * C * C * 9 * P * B * S * S
print("B=%d S=%d: %.3f ms per call (with the two |max| passes), %.0f TFLOP/s fp32-equivalent, %.2f of the fp16 peak as products"
      % (B, S, ms, fl / ms / 1e9, 3 * fl / ms / 1e9 / 2500))
