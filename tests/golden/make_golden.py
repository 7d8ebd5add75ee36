"""Generates tests/golden/ref_*.npz by RUNNING THE REFERENCE'S OWN PYTHON on CPU (build container only).

    python tests/golden/make_golden.py            # needs /root/reference; never runs on the GPU box

What is the reference's own code here and what is not:
  * stage A -- imported by file path with NO stubs: graphs/layers/{P_block_v2,wavelet_forward_v2,
    wavelet_inverse_v2,masked_conv2d}.py, utils/{bound_ops,parametrizers}.py.
  * stage B -- graphs/layers/lifting_dwt_nets.py needs three module NAMES that are absent from the image:
    ``compressai.layers.GDN`` (bound to the reference's own vendored copy graphs/layers/gdn.py, whose
    ``compressai.ops.parametrizers`` import is bound to the reference's own utils/parametrizers.py),
    ``pytorch_wavelets`` (placeholder names, never called: the CDF97 path is pinned by pywt instead) and the package
    ``__init__`` auto-importers (bypassed by pre-registering bare package modules).
  * stage C -- graphs/models/LiftingBasedDWT_net.py additionally needs ``compressai.entropy_models`` and
    ``compressai.ans``.  The two leaf classes (EntropyBottleneck, GaussianConditional) are provided by thin nn.Modules
    around oracle/entropy.py's restatement of compressai 1.2.1 -> the *wiring* (context CNNs, masks, regrouping,
    upsampling, level order, quantisation quirks) in these fixtures is the reference's, the leaf likelihood
    arithmetic is the restatement ("parity unpinned" for the leaf ops, see oracle/entropy.py).
Weights come from oracle.weights.fill_by_name (crc32(key)-seeded), so tests can regenerate them bit-for-bit; a
checksum of the weights is stored with each fixture.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle import entropy as oent            # noqa: E402
from oracle import model as omodel            # noqa: E402
from oracle import weights as oweights        # noqa: E402

torch.set_num_threads(8)


class Cfg(dict):
    __getattr__ = dict.__getitem__


def _bare_package(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def setup_reference_imports():
    for pkg in ("graphs", "graphs/layers", "graphs/models", "graphs/losses", "utils", "agents"):
        _bare_package(pkg.replace("/", "."), os.path.join(REF, pkg))
    import matplotlib
    matplotlib.use("Agg")
    # --- name stubs -------------------------------------------------------------------------------------------
    comp = types.ModuleType("compressai")
    comp.__path__ = []
    sys.modules["compressai"] = comp
    ops = types.ModuleType("compressai.ops")
    ops.__path__ = []
    sys.modules["compressai.ops"] = ops
    sys.modules["compressai.ops.parametrizers"] = importlib.import_module("utils.parametrizers")
    gdn_mod = importlib.import_module("graphs.layers.gdn")
    layers = types.ModuleType("compressai.layers")
    layers.GDN = gdn_mod.GDN
    layers.GDN1 = gdn_mod.GDN1
    sys.modules["compressai.layers"] = layers
    pw = types.ModuleType("pytorch_wavelets")
    pw.DWTForward = pw.DWTInverse = None
    sys.modules["pytorch_wavelets"] = pw
    vis = types.ModuleType("visdom")
    vis.Visdom = None
    sys.modules["visdom"] = vis
    ans = types.ModuleType("compressai.ans")
    ans.BufferedRansEncoder = ans.RansDecoder = None
    sys.modules["compressai.ans"] = ans
    em = types.ModuleType("compressai.entropy_models")
    em.EntropyBottleneck = StubEntropyBottleneck
    em.GaussianConditional = StubGaussianConditional
    sys.modules["compressai.entropy_models"] = em


class StubGaussianConditional(nn.Module):
    """Leaf stand-in: oracle/entropy.py restatement of compressai 1.2.1 GaussianConditional (see module docstring)."""

    def __init__(self, scale_table=None, scale_bound=0.11, **kw):
        super().__init__()
        assert abs(scale_bound - oent.SCALE_BOUND) < 1e-12

    def quantize(self, inputs, mode, means=None):
        return oent.quantize(inputs, mode, means)

    def forward(self, inputs, scales, means=None, training=None):
        training = self.training if training is None else training
        return oent.gaussian_conditional_forward(inputs, scales, means, training)


class StubEntropyBottleneck(nn.Module):
    """Leaf stand-in: oracle/entropy.py restatement of compressai 1.2.1 EntropyBottleneck."""

    def __init__(self, channels, **kw):
        super().__init__()
        for k, v in oent.eb_init_state(int(channels)).items():
            if k == "target":
                self.register_buffer(k, v)
            else:
                self.register_parameter(k, nn.Parameter(v))

    def forward(self, x, training=None):
        training = self.training if training is None else training
        sd = {k: v for k, v in self.state_dict().items()}
        return oent.entropy_bottleneck_forward(x, sd, "", training)


def load_by_name(module, prefix=""):
    """Fill a reference module from oracle.weights.fill_value keyed by the *unique* (non-aliased) names."""
    own = module.state_dict()
    new = {}
    for k, v in own.items():
        if "waveletForward." in k or "waveletInverse." in k:
            continue   # aliases of P_blocks / U_blocks / preProcessingList / nh / nl (shared modules)
        new[k] = oweights.fill_value(prefix + k, v).to(v.dtype).reshape(v.shape)
    module.load_state_dict(new, strict=False)
    return {prefix + k: v for k, v in new.items()}


def checksum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, len(out), "arrays")


def seeded(shape, seed, smooth=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(shape, generator=g)
    if smooth:
        x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode="replicate"), 5, stride=1)
    return x


def main():
    setup_reference_imports()
    base = Cfg(omodel.DEFAULT_CFG)

    # ------------------------------------------------------------------ stage A: no stubs
    from graphs.layers.P_block_v2 import P_block_v2
    from graphs.layers.masked_conv2d import MaskedConv2d
    from utils.bound_ops import LowerBound
    from utils.parametrizers import NonNegativeParametrizer

    for k in (3, 5):
        blk = P_block_v2(1, 1, k, 16)
        sd = load_by_name(blk, "P_blocks.0.")
        x = seeded((2, 1, 12, 20), 7) - 0.5
        save("ref_pblock_k%d" % k, x=x, y=blk(x), wsum=checksum(sd))
    blk = P_block_v2(0, 1, 3, 16)     # linearity_flag != 1
    sd = load_by_name(blk, "P_blocks.0.")
    x = seeded((1, 1, 8, 8), 8) - 0.5
    save("ref_pblock_linear", x=x, y=blk(x), wsum=checksum(sd))

    for mt in ("A", "B"):
        for k in (3, 5):
            mc = MaskedConv2d(mt, 3, 6, k, 1, k // 2, groups=3)
            sd = load_by_name(mc, "csc_list.0.")
            x = seeded((1, 3, 9, 11), 9) - 0.5
            save("ref_maskedconv_%s%d" % (mt, k), x=x, y=mc(x), mask=mc.mask, wsum=checksum(sd))

    x = torch.tensor([-1.0, 0.0, 0.11, 0.5, 2.0, 1e-10], requires_grad=True)
    lb = LowerBound(0.11)
    y = lb(x)
    gup = torch.tensor([1.0, -1.0, 2.0, -0.5, 1.0, -3.0])
    y.backward(gup)
    save("ref_lower_bound", x=x, y=y, gup=gup, gx=x.grad)
    x = torch.tensor([-1.0, 0.0, 0.5, 2.0, 1e-3, 3e-6], requires_grad=True)
    npz = NonNegativeParametrizer(minimum=1e-6)
    y = npz(x)
    y.backward(torch.ones_like(x))
    save("ref_nonneg_param", x=x, y=y, gx=x.grad, init=npz.init(torch.tensor([0.0, 0.1, 1.0, 4.0])))

    # ------------------------------------------------------------------ stage B: lifting auto-encoder
    from graphs.layers.lifting_dwt_nets import LiftingBasedNeuralWaveletv4
    from graphs.layers.gdn import GDN

    g = GDN(6)
    gi = GDN(6, inverse=True)
    sdg = load_by_name(g, "Yl_ae.ae_down.1.")
    load_by_name(gi, "Yl_ae.ae_down.1.")
    x = (seeded((1, 6, 5, 7), 10) - 0.5) * 4
    xg = x.clone().requires_grad_(True)
    yg = g(xg)
    yg.sum().backward()
    save("ref_gdn", x=x, y=yg, y_inv=gi(x), gx=xg.grad, gbeta=g.beta.grad, ggamma=g.gamma.grad, wsum=checksum(sdg))

    def lifting_case(name, shape, seed, **over):
        cfg = Cfg(base)
        cfg.update(over)
        net = LiftingBasedNeuralWaveletv4(cfg).eval()
        sd = load_by_name(net)
        x = seeded(shape, seed, smooth=True) - 0.5
        with torch.no_grad():
            LL, LH, HL, HH = net.waveletForward[0].one_level_lifting(x)
            rec1 = net.waveletInverse[0].one_level_lifting(LL, LH, HL, HH)
            out_xe, out_xo = net.encode(x)
            xr = net.decode(out_xe, out_xo)
        arrs = dict(x=x, LL=LL, LH=LH, HL=HL, HH=HH, rec1=rec1, out_xe=out_xe, xr=xr, wsum=checksum(sd),
                    cfg=json.dumps(dict(cfg)))
        for i, t in enumerate(out_xo):
            arrs["out_xo%d" % i] = t
        save(name, **arrs)

    lifting_case("ref_lifting_L2_k5", (1, 1, 32, 32), 11, dwtlevels=2)
    lifting_case("ref_lifting_L3_k3_rect", (2, 1, 32, 48), 12, dwtlevels=3, filtersize=3)
    lifting_case("ref_lifting_L2_scale_berk", (1, 1, 16, 32), 13, dwtlevels=2, scale=1,
                 autoencoder="SubbandAutoEncoderBerk")
    lifting_case("ref_lifting_L2_different", (1, 1, 16, 16), 14, dwtlevels=2, block_property="different")
    lifting_case("ref_lifting_L2_linear", (1, 1, 16, 16), 15, dwtlevels=2, linearity_flag=0, filtersize=3)

    # skip filters on an impulse and a ramp: border behaviour (zero padding)
    cfg = Cfg(base)
    cfg.update(dwtlevels=1)
    net = LiftingBasedNeuralWaveletv4(cfg).eval()
    imp = torch.zeros(1, 1, 8, 3)
    imp[0, 0, 0, 0] = 1.0
    imp[0, 0, 7, 1] = 1.0
    imp[0, 0, 3, 2] = 1.0
    ramp = torch.arange(8.0).view(1, 1, 8, 1).repeat(1, 1, 1, 3)
    with torch.no_grad():
        arrs = {"imp": imp, "ramp": ramp}
        for j in range(4):
            arrs["imp%d" % j] = net.preProcessingList[j](imp)
            arrs["ramp%d" % j] = net.preProcessingList[j](ramp)
            arrs["w%d" % j] = net.preProcessingList[j].weight
    save("ref_skip_filters", **arrs)

    # ------------------------------------------------------------------ stage C: entropy layers + wrapper + loss
    from graphs.models.LiftingBasedDWT_net import LiftingBasedDWTNetWrapper
    from graphs.losses.rate_dist import TrainRDLoss

    def wrapper_case(name, shape, seed, **over):
        cfg = Cfg(base)
        cfg.update(over)
        net = LiftingBasedDWTNetWrapper(cfg).eval()
        sd = load_by_name(net)
        x = seeded(shape, seed, smooth=True)          # RGB in [0,1]
        y = omodel.rgb2ycbcr(x) - omodel._YSHIFT       # agent colour maths (compressai.transforms absent)
        with torch.no_grad():
            # per-plane intermediates from the reference modules
            arrs = dict(x=x, y=y, wsum=checksum(sd), cfg=json.dumps(dict(cfg)))
            for c, m in enumerate((net.model0, net.model1, net.model2)):
                oxe, oxo = m.autoencoder.encode(y[:, c:c + 1])
                si_xe, si_xo, qxe, qxo = m.entropymodel(oxe, oxo)
                arrs["p%d_out_xe" % c] = oxe
                arrs["p%d_si_xe" % c] = si_xe
                arrs["p%d_q_xe" % c] = qxe
                for i in range(len(oxo)):
                    arrs["p%d_out_xo%d" % (c, i)] = oxo[i]
                    arrs["p%d_si_xo%d" % (c, i)] = si_xo[i]
                    arrs["p%d_q_xo%d" % (c, i)] = qxo[i]
            yhat, si_xe, si_xo = net(y)
            xhat = omodel.ycbcr2rgb(yhat + omodel._YSHIFT) - 0.5
            loss, mse, r1, r2 = TrainRDLoss(cfg.lambda_).forward3(x - 0.5, xhat, si_xe, si_xo)
            arrs.update(yhat=yhat, xhat=xhat, loss=loss, mse=mse, rate1=r1, rate2=r2,
                        n_si_xo=len(si_xo))
        save(name, **arrs)

    wrapper_case("ref_wrapper_cond2_L3", (1, 3, 64, 64), 21, dwtlevels=3)
    wrapper_case("ref_wrapper_ezwt_L3", (1, 3, 32, 64), 22, dwtlevels=3, entropy_layer="onlyEZWT")
    wrapper_case("ref_wrapper_fact_L2", (2, 3, 32, 32), 23, dwtlevels=2, entropy_layer="factorized")
    wrapper_case("ref_wrapper_cond2_berk_L2", (1, 3, 32, 32), 24, dwtlevels=2, autoencoder="SubbandAutoEncoderBerk")
    # DWTConditioned2EntropyLayerZTBlock.forward allocates with `.cuda()` unconditionally (LiftingBasedDWT_net.py:717-718);
    # the build container has no GPU, so Tensor.cuda is made a no-op for this one call (the reference source is unchanged)
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        wrapper_case("ref_wrapper_ztblock_L3", (1, 3, 32, 32), 25, dwtlevels=3,
                     entropy_layer="DWTConditioned2EntropyLayerZTBlock")
    finally:
        torch.Tensor.cuda = _cuda


if __name__ == "__main__":
    main()
