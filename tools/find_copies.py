#!/usr/bin/env python
"""Diagnostic: which torch (ATen) device ops run inside one eval step of bench.py's workload besides the C-ABI launches --
in particular the device-to-device copies rocprofv3 shows as __amd_rocclr_copyBuffer.  Counts ATen ops per
step through a TorchDispatchMode and prints the Python call sites of the copying ones.   python tools/find_copies.py"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = collections.Counter()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        self.ops[name] += 1
        if any(k in name for k in ("copy", "clone", "cat", "stack", "contiguous", "_to_copy", "index", "zeros", "fill", "zero_")):
            fr = [f for f in traceback.extract_stack()[:-1] if "imagecompression" in f.filename or "bench" in f.filename]
            site = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[-3:][::-1])
            self.sites[(name, site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    import bench
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net import rate_planes
    dev = torch.device("cuda", 0)
    c = dict(bench.CONFIGS[int(os.environ.get("CFG", "2"))])
    net, sd, cfg = bench.build_model(c, dev)
    nets = net.nets()
    x = torch.rand(c["batch"], 3, c["H"], c["W"], device=dev)
    acc = torch.zeros(1, dtype=torch.float64, device=dev)

    def step():
        with torch.no_grad():
            y = ops.rgb_to_ycc(x)
            si_xe, si_xo = rate_planes(nets, y, False)
            ops.sum_into(si_xe, acc)
            for t in si_xo:
                ops.sum_into(t, acc)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with Counter() as cnt:
        step()
    torch.cuda.synchronize()
    print("ATen ops in one step: %d" % sum(cnt.ops.values()))
    for k, v in cnt.ops.most_common(25):
        print("  %4d  %s" % (v, k))
    print("call sites of the copying / allocating ones:")
    for (name, site), v in cnt.sites.most_common(40):
        print("  %4d  %-28s %s" % (v, name, site))


if __name__ == "__main__":
    main()
