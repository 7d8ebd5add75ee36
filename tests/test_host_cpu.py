"""CPU (no GPU): the C-ABI library loads and exports every symbol include/lldwt.h declares; the host-side mirror has the
reference's module API and state_dict layout; the product path has no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

from oracle import model, weights

PKG = "imagecompressionlearnedliftingandlearnedtreebasedmodels_amd"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(REPO, "include", "lldwt.h")).read()
    declared = set(re.findall(r"\b(lldwt_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"lldwt_view", "lldwt_conv_desc"}
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), name                      # exported by the .so
        assert name in _lib.SIGNATURES, name                 # bound with a prototype
    assert lib.lldwt_version() >= 100
    assert lib.lldwt_pblock_packed_floats(16, 5) > 2 * (16 * 16 * 25) * 2


def test_no_gpu_calls_fail_loudly():
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd._lib import LLDWTError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(LLDWTError):
        ops.quantize(torch.zeros(4))          # host tensor: no silent CPU path


def test_missing_library_is_an_error(tmp_path, monkeypatch):
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.LLDWTError):
        _lib.load()


def test_product_never_imports_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, PKG)):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                s = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, re.M) or "oracle/" in s:
                    bad.append(f)
    assert not bad, bad


@pytest.mark.parametrize("ent", ["factorized", "onlyEZWT", "conditioned2ZTsepSubbands", "DWTConditioned2EntropyLayerZTBlock"])
@pytest.mark.parametrize("ae", ["SubbandAutoEncoder", "SubbandAutoEncoderBerk"])
@pytest.mark.parametrize("nt", ["LiftingBasedNeuralWaveletv4", "CDF97"])
def test_state_dict_layout_matches_reference(ent, ae, nt):
    """Key names and shapes equal the reference modules' (oracle.weights templates were checked key-for-key against the
    reference's own state_dict in tests/golden/make_golden.py)."""
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net import \
        LiftingBasedDWTNetWrapper
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    cfg = make_config(dwtlevels=3, entropy_layer=ent, autoencoder=ae, netType=nt)
    net = LiftingBasedDWTNetWrapper(cfg)
    sd = net.state_dict()
    tpl = weights.wrapper_template(dict(cfg))
    compressai_buffers = ("_offset", "_quantized_cdf", "_cdf_length", "scale_table", "scale_bound", "lower_bound", "_reparam")
    mine = {k: v for k, v in sd.items() if "waveletForward" not in k and "waveletInverse" not in k}
    assert not [k for k in tpl if k not in mine]
    assert not [k for k in mine if k not in tpl and not any(t in k for t in compressai_buffers)]
    assert not [k for k in tpl if tuple(mine[k].shape) != tuple(tpl[k].shape)]
    if nt != "CDF97":   # aliases of the shared blocks, SURVEY 8b
        assert "model0.autoencoder.waveletForward.0.P.0.conv1.weight" in sd
        assert "model0.autoencoder.waveletInverse.2.convBlock.3.weight" in sd
        a = net.model0.autoencoder
        assert a.waveletForward[0].P[0] is a.P_blocks[0]
    net.load_state_dict({k: v for k, v in weights.fill_by_name(tpl).items()}, strict=False)


def test_config_and_agent_api_surface():
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.agents.liftingDWT_agent import (
        LiftingBasedDWTAgent, configure_optimizers)
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    for m in ("run", "finalize", "train_one_epoch", "validate", "test", "load_checkpoint", "save_checkpoint"):
        assert callable(getattr(LiftingBasedDWTAgent, m))
    cfg = make_config(dwtlevels=2)
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net import \
        LiftingBasedDWTNetWrapper
    net = LiftingBasedDWTNetWrapper(cfg)
    opt = configure_optimizers(net, 1e-4)
    n = sum(p.numel() for p in net.parameters() if p.requires_grad)
    assert sum(p.numel() for g in opt.param_groups for p in g["params"]) == n
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            LiftingBasedDWTAgent(cfg)


def test_masked_conv_mask_matches_reference_fixture():
    from helpers import load_golden
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.layers.masked_conv2d import MaskedConv2d
    for mt in "AB":
        for k in (3, 5):
            m = MaskedConv2d(mt, 3, 6, k, 1, k // 2, groups=3)
            assert torch.equal(m.mask, load_golden("ref_maskedconv_%s%d" % (mt, k))["mask"])
            assert bin(m.tap_bits()).count("1") == int(m.mask[0, 0].sum())


def test_csc_fold_algebra_matches_the_unfused_layers():
    """Host logic of the folded context (graphs/models/LiftingBasedDWT_net.py `_fold_csc_train`): cgp layer 0 applied to
    [plc_g | csc_g] (LiftingBasedDWT_net.py:353-359, csc = masked 5x5 type-A grouped conv of the quantised subband) equals
    the folded layer applied to [plc_g | 12 live taps of the subband] -- values and the gradients that autograd carries
    back to W0, the csc weight / bias and the subband.  Pure tensor algebra: runs on the CPU."""
    import torch.nn.functional as F
    from torch import nn
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.layers.masked_conv2d import MaskedConv2d
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net import _fold_csc_train
    torch.manual_seed(3)
    P, B, G, cpl, cc, c1, h, w = 2, 2, 3, 5, 4, 7, 6, 9
    cg = [nn.Sequential(nn.Conv2d(G * (cpl + cc), G * c1, 1, groups=G)).double() for _ in range(P)]
    cs = [MaskedConv2d("A", G, G * cc, 5, padding=2, groups=G).double() for _ in range(P)]
    xq = torch.randn(P, B, G, h, w, dtype=torch.float64, requires_grad=True)
    plc = torch.randn(P, B, G * cpl, h, w, dtype=torch.float64)
    gout = torch.randn(P, B, G * c1, h, w, dtype=torch.float64)

    def grads():
        g = [xq.grad.clone()] + [p.grad.clone() for m in cg for p in m.parameters()] + \
            [p.grad.clone() for m in cs for p in m.parameters()]
        xq.grad = None
        for m in list(cg) + list(cs):
            m.zero_grad()
        return g

    # reference order: csc conv, regroup (p0,c0,p1,c1,p2,c2), grouped 1x1 conv
    outs = []
    for p in range(P):
        csc = F.conv2d(xq[p], cs[p].weight * cs[p].mask, cs[p].bias, padding=2, groups=G)   # the module itself is HIP-only
        pl, c_ = plc[p].chunk(G, 1), csc.chunk(G, 1)
        cat = torch.cat([t for g_ in range(G) for t in (pl[g_], c_[g_])], 1)
        outs.append(cg[p][0](cat))
    ref = torch.stack(outs, 0)
    (ref * gout).sum().backward()
    gref = grads()
    # folded
    w0f, b0f, patches = _fold_csc_train(cg, cs, xq)
    pl, pa = plc.chunk(G, 2), patches.chunk(G, 2)
    t = torch.cat([z for g_ in range(G) for z in (pl[g_], pa[g_])], 2)
    got = torch.stack([F.conv2d(t[p], w0f[p], b0f[p], groups=G) for p in range(P)], 0)
    assert float((got - ref).detach().abs().max()) < 1e-10
    (got * gout).sum().backward()
    gfold = grads()
    for a, b in zip(gfold, gref):
        assert float((a - b).abs().max()) < 1e-9


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("scale", [False, True])
def test_lifting_program_layout_host_side(inverse, scale):
    """lldwt_lifting_program is host code (no GPU): per level 12 lifting steps (+ 6 gain ops with scale == 1,
    wavelet_forward_v2.py:76-80 / wavelet_inverse_v2.py:70-74); every op owns a disjoint slice of the saved buffer -- a
    lifting step (2 + 3 C) floats per element, a gain op one float per element (its input, which the backward needs) --
    and the slices tile [0, saved_floats) in program order."""
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    Z, H, W, L, C_ = 6, 64, 96, 3, 16
    prog, total = ops.lifting_program(Z, H, W, L, False, 0, inverse, C_, scale)
    steps = [o for o in prog if o.kind == 0]
    gains = [o for o in prog if o.kind != 0]
    assert len(steps) == 12 * L and len(gains) == (6 * L if scale else 0)
    assert {o.kind for o in gains} == (set() if not scale else ({3, 4} if inverse else {1, 2}))
    off = 0
    for o in prog:
        assert o.saved_off == off
        n = Z * o.h * o.w
        off += n * (2 + 3 * C_) if o.kind == 0 else n
    assert off == total
    # the gain ops of the forward transform work in place; the inverse's first ones write scaled copies elsewhere
    for o in gains:
        same = (o.buf_src, o.off_src) == (o.buf_dout, o.off_dout)
        assert same or inverse
    # without gains the program and its saved size are what they were before gain ops kept anything
    prog0, total0 = ops.lifting_program(Z, H, W, L, False, 0, inverse, C_, False)
    assert total0 == sum(Z * o.h * o.w * (2 + 3 * C_) for o in prog0)


def test_wavefront_step_offsets_match_the_pixel_order():
    """graphs/models/entropy_coding._step_offsets (what code_tree_level passes to lldwt_cgp16_wavefront_step as `off`) against
    wavefront(): step t holds the pixels x + s*y == t, rows ascending -- the order the streams are written in."""
    import numpy as np
    import torch
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models import entropy_coding as ec
    for H, W, s in ((4, 4, 2), (5, 9, 3), (16, 7, 3), (1, 6, 3), (8, 8, 2)):
        hs, ws, starts = ec.wavefront(H, W, s, torch.device("cpu"))
        assert ec._step_offsets(H, W, s) == starts
        t = (ws + s * hs).numpy()
        assert (np.diff(t) >= 0).all() and starts[-1] == H * W and len(starts) == W + s * (H - 1) + 1
        for k in range(len(starts) - 1):
            seg = slice(starts[k], starts[k + 1])
            assert (t[seg] == k).all() and (np.diff(hs[seg].numpy()) > 0).all()       # rows ascending inside a step
            # the kernel's own enumeration of the step: rows y0 .. y0 + n - 1, x = t - s*y inside [0, W)
            lo = max(0, -(-(k - (W - 1)) // s)) if k - (W - 1) > 0 else 0
            hi = min(H - 1, k // s)
            n = hi - lo + 1 if hi >= lo else 0
            assert n == starts[k + 1] - starts[k]
            if n:
                assert int(hs[starts[k]]) == lo


def test_precision_names_and_traffic_hashes(monkeypatch):
    """LLDWT_PRECISION is validated at load (names -> codes of lldwt_set_precision); bench.py's traffic figures are tied to the
    git blob hash of the kernel sources they were measured on."""
    import hashlib
    import json
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import _lib
    assert _lib.PRECISIONS == {"f16x3": 0, "fp16": 1, "bf16": 2}
    import bench
    h = bench.source_hashes()
    src = os.path.join(REPO, PKG, "csrc", "lifting_f16.hip")
    b = open(src, "rb").read()
    assert h["lifting_f16.hip"] == hashlib.sha1(b"blob %d\0" % len(b) + b).hexdigest()
    assert bench.traffic_is_current({"source_hashes": dict(h)}, ("lifting_f16.hip", "split_f16.h"))
    stale = dict(h, **{"lifting_f16.hip": "0" * 40})
    assert not bench.traffic_is_current({"source_hashes": stale}, ("lifting_f16.hip",))
    assert not bench.traffic_is_current({}, ("conv_f16x3.hip",))
    tj = json.load(open(os.path.join(REPO, "profiles", "traffic_current.json")))
    assert set(tj["source_hashes"]) >= {"conv_f16x3.hip", "lifting_f16.hip"}
