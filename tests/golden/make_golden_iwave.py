"""Generates tests/golden/ref_iwave.npz by RUNNING THE REFERENCE'S OWN PostProcessingiWave
(graphs/layers/post_processing_networks.py:54-77, with PostProcessResidual :39-52) on CPU -- build container only.

    python tests/golden/make_golden_iwave.py        # needs /root/reference; never runs on the GPU box

Imported through make_golden.setup_reference_imports (bare package modules, no stand-in for anything this file calls: the
module and its basic_block import are plain torch).  Weights by name from oracle.weights.fill_value (crc32(key)-seeded),
with a livelier scale on the residual blocks than the reference's 0.01 initialisation (which makes them ~identity).
Stores the input, the output, and the gradients of sum(y^2) w.r.t. the input and four representative parameters."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402


def main():
    mg.setup_reference_imports()
    from graphs.layers.post_processing_networks import PostProcessingiWave
    cfg = mg.Cfg(clrch=1, resnetlevel=2)
    net = PostProcessingiWave(cfg)
    sd = mg.load_by_name(net, "iwave.")
    with torch.no_grad():
        for k, p in net.named_parameters():
            p.mul_(0.25)                               # fill_value's scale is sized for the codec's layers
    x = (mg.seeded((2, 3, 24, 40), 31, smooth=True)).clone().requires_grad_(True)
    y = net(x)
    (y ** 2).sum().backward()
    g = dict(net.named_parameters())
    mg.save("ref_iwave", x=x.detach(), y=y, gx=x.grad, wsum=mg.checksum({k: v.detach() for k, v in g.items()}),
            g_convFilter_weight=g["convFilter.weight"].grad, g_res1_conv2_weight=g["resNetList.1.resNet.2.weight"].grad,
            g_res0_conv0_bias=g["resNetList.0.resNet.0.bias"].grad, g_outputConvFilter_weight=g["outputConvFilter.weight"].grad)


if __name__ == "__main__":
    main()
