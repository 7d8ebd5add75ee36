"""What the memory system gives a trivial kernel with the Gaussian rate kernel's traffic mix (three reads + one write of 4 B per
element, torch.addcmul) at its two bench sizes -- HIP-graph replay of 20 back-to-back calls, like bench.hbm_kernels."""
import json
import torch

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e-3


for B in (8, 96):
    n = 3 * B * 3 * 256 * 256
    a, b, c, d = (torch.rand(n, device=dev) for _ in range(4))
    for name, fn, nb in (("addcmul 3r+1w", lambda: torch.addcmul(a, b, c, out=d), 16 * n),
                         ("add 2r+1w", lambda: torch.add(a, b, out=d), 12 * n),
                         ("copy 1r+1w", lambda: d.copy_(a), 8 * n),
                         ("mul_ scalar 1r+1w in place", lambda: d.mul_(1.0001), 8 * n)):
        t = timeit(fn)
        print(json.dumps({"B": B, "op": name, "MB": nb / 1e6, "us": round(t * 1e6, 2), "TB/s": round(nb / t / 1e12, 3),
                          "frac_of_8TBs": round(nb / t / 8e12, 3)}), flush=True)
