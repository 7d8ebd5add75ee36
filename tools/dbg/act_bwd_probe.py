import torch, sys, os
sys.path.insert(0, os.getcwd())
from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
dev = torch.device("cuda:0")
for shape in ((3, 8, 243, 256, 256), (3, 8, 243, 128, 128), (3, 8, 16, 256, 256)):
    dy = torch.randn(*shape, device=dev); y = torch.randn(*shape, device=dev)
    for act in (ops.ACT_LRELU,):
        r = ops.act_bwd(dy, y, act)
        ref = torch.where(y > 0, dy, 0.01 * dy)
        assert torch.equal(r, ref)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.act_bwd(dy, y, act)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(shape, "us %.1f" % (t * 1e6), "TB/s %.2f" % (dy.numel() * 12 / t / 1e12), flush=True)
    del dy, y, r, ref
