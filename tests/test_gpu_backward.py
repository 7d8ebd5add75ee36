"""GPU parity of the hand-written BACKWARD kernels vs torch-CPU autograd over the oracle's maths."""
import pytest
import torch
import torch.nn.functional as F

from helpers import maxdiff
from oracle import entropy

pytestmark = pytest.mark.gpu


def _mods():
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import autograd as ag
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    import gpu_util
    return ag, ops, gpu_util


BWD_CASES = [
    # cin, cout, K, groups, act, upsample, masktype
    (16, 16, 5, 1, 1, False, None),        # lifting conv2 (tanh)
    (3, 243, 3, 1, 2, True, None),         # plc first layer on the upsampled parent (LeakyReLU)
    (243, 243, 3, 1, 0, False, None),      # plc second layer
    (3, 243, 5, 3, 0, False, "A"),         # csc masked 5x5 grouped
    (243, 81, 3, 3, 2, False, "B"),        # masked 3x3 B grouped
    (486, 162, 1, 3, 2, False, None),      # cgp 1x1 grouped (64 x 192 tile of the 1x1 weight-gradient GEMM)
    (486, 486, 1, 3, 2, False, None),      # cgp first layer: 162 x 162 per group = two 96-row blocks
    (54, 6, 1, 3, 0, False, None),
    (1, 32, 1, 1, 1, False, None),         # subband AE first layer
    (96, 3, 1, 3, 0, False, None),         # subband AE last layer (grouped 32 -> 1)
]


@pytest.mark.parametrize("case", BWD_CASES)
def test_conv_backward(case):
    ag, ops, gu = _mods()
    cin, cout, K, groups, act, up, mt = case
    P, B, h, w = 2, 3, 12, 20
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    hi, wi = (h // 2, w // 2) if up else (h, w)
    x = torch.rand(P, B, cin, hi, wi, generator=g) - 0.5
    wt = (torch.rand(P, cout, cin // groups, K, K, generator=g) - 0.5) * 0.3
    b = torch.rand(P, cout, generator=g) - 0.5
    gy = torch.rand(P, B, cout, h, w, generator=g) - 0.5
    mask_bits, m = None, None
    if mt:
        m = entropy.conv_mask((cout, cin // groups, K, K), mt)
        wt = wt * m
        mask_bits = int(sum(1 << t for t in range(K * K) if m[0, 0].flatten()[t] > 0))
    xd, wd, bd = (gu.dev(t).requires_grad_(True) for t in (x, wt, b))
    y = ag.conv(xd, wd, bd, K, groups=groups, act=act, upsample2=up, tap_mask=mask_bits)
    y.backward(gu.dev(gy))
    for p in range(P):
        xr, wr, br = (t[p].clone().requires_grad_(True) for t in (x, wt, b))
        xi = entropy.upsample2(xr) if up else xr
        ref = F.conv2d(xi, wr, br, padding=K // 2, groups=groups)
        ref = torch.tanh(ref) if act == 1 else (F.leaky_relu(ref, 0.01) if act == 2 else ref)
        assert maxdiff(y[p].detach().cpu(), ref) < 2e-5
        ref.backward(gy[p])
        scale = max(1.0, float(wr.grad.abs().max()))
        assert maxdiff(xd.grad[p].cpu(), xr.grad) < 1e-4, case
        gw_ref = wr.grad * m if m is not None else wr.grad        # dead taps are not computed (re-zeroed every forward)
        assert maxdiff(wd.grad[p].cpu(), gw_ref) < 2e-4 * scale, case
        assert maxdiff(bd.grad[p].cpu(), br.grad) < 2e-4 * max(1.0, float(br.grad.abs().max())), case


@pytest.mark.parametrize("hw", [(21, 37), (72, 136)])
def test_conv_backward_residual_and_large(hw):
    """Residual branch gradient + a spatial size that is not a multiple of the 8x16 wgrad chunk; 72x136 has interior
    tiles of every conv-engine tile shape and several weight-gradient chunks per K slice."""
    ag, ops, gu = _mods()
    g = torch.Generator().manual_seed(5)
    P, B, C, K = 1, 2, 16, 5
    h, w = hw
    x = torch.rand(P, B, C, h, w, generator=g) - 0.5
    wt = (torch.rand(P, C, C, K, K, generator=g) - 0.5) * 0.2
    b = torch.rand(P, C, generator=g) - 0.5
    res = torch.rand(P, B, C, h, w, generator=g) - 0.5
    gy = torch.rand(P, B, C, h, w, generator=g) - 0.5
    xd, wd, bd, rd = (gu.dev(t).requires_grad_(True) for t in (x, wt, b, res))
    y = ag.conv(xd, wd, bd, K, residual=rd)
    y.backward(gu.dev(gy))
    xr, wr, br, rr = (t[0].clone().requires_grad_(True) for t in (x, wt, b, res))
    ref = F.conv2d(xr, wr, br, padding=2) + rr
    ref.backward(gy[0])
    assert maxdiff(xd.grad[0].cpu(), xr.grad) < 1e-4
    assert maxdiff(wd.grad[0].cpu(), wr.grad) < 2e-4 * max(1.0, float(wr.grad.abs().max()))
    assert maxdiff(rd.grad[0].cpu(), rr.grad) < 1e-6


@pytest.mark.parametrize("cin,cout", [(16, 16), (1, 16), (16, 1)])
@pytest.mark.parametrize("K", [3, 5])
@pytest.mark.parametrize("swap,geom", [(False, (2, 3, 29, 43)), (True, (2, 3, 29, 43)),
                                       # many 8x16 chunks per K slice, several slices per plane
                                       (False, (1, 2, 136, 200)), (True, (1, 1, 128, 96))])
def test_wgrad_lifting_shapes(cin, cout, K, swap, geom):
    """The dedicated P-block weight-gradient kernels (16 -> 16 tap-tiled, 1 <-> 16 thin): ragged spatial size, several
    K slices, alpha scaling, accumulation into a non-zero dW, and the (kh,kw)-swapped output of the row passes."""
    ag, ops, gu = _mods()
    g = torch.Generator().manual_seed(100 * cin + 10 * cout + K)
    P, B, h, w = geom
    x = torch.rand(P, B, cin, h, w, generator=g) - 0.5
    gy = torch.rand(P, B, cout, h, w, generator=g) - 0.5
    dw0 = torch.rand(P, cout, cin, K, K, generator=g) - 0.5
    db0 = torch.rand(P, cout, generator=g) - 0.5
    alpha = -0.1
    dw, db = gu.dev(dw0.clone()), gu.dev(db0.clone())
    ops.conv2d_wgrad(gu.dev(x), gu.dev(gy), (P, cout, cin, K, K), K, dw=dw, db=db, alpha=alpha, swap_hw=swap)
    for p in range(P):
        wr = torch.zeros(cout, cin, K, K, requires_grad=True)
        br = torch.zeros(cout, requires_grad=True)
        F.conv2d(x[p], wr, br, padding=K // 2).backward(gy[p])
        gw = wr.grad.transpose(2, 3) if swap else wr.grad
        scale = max(1.0, float(gw.abs().max()))
        assert maxdiff(dw[p].cpu(), dw0[p] + alpha * gw) < 2e-5 * scale, (cin, cout, K, swap)
        assert maxdiff(db[p].cpu(), db0[p] + alpha * br.grad) < 2e-5 * max(1.0, float(br.grad.abs().max()))


@pytest.mark.parametrize("geom,swap", [((2, 3, 29, 44), False), ((1, 2, 136, 200), True), ((3, 1, 64, 256), False), ((1, 1, 7, 12), True)])
def test_wgrad16_split_fp16(geom, swap):
    """k_wgrad16_f16x3 (the 16 -> 16 5x5 weight gradient of conv2 / conv3 on the fp16 matrix cores): tanh-range inputs, gradients of
    very different magnitudes per plane (the per-plane power-of-two scale), ragged sizes (rows not a multiple of the 8-row chunk,
    widths not a multiple of 32), several chunks per workgroup, alpha, accumulation into a non-zero dW and the (kh,kw)-swapped
    output of the row passes -- vs torch autograd, fp32-level bar (2e-5 of the largest entry)."""
    ag, ops, gu = _mods()
    P, B, h, w = geom
    g = torch.Generator().manual_seed(7 * h + w)
    x = torch.tanh(torch.randn(P, B, 16, h, w, generator=g) * 2.0)
    gy = (torch.rand(P, B, 16, h, w, generator=g) - 0.5) * torch.tensor([3e-4, 1.0, 250.0])[:P].reshape(P, 1, 1, 1, 1)
    dw0 = torch.rand(P, 16, 16, 5, 5, generator=g) - 0.5
    db0 = torch.rand(P, 16, generator=g) - 0.5
    alpha = -0.1
    dw, db = ops.wgrad16_f16x3(gu.dev(x), gu.dev(gy), dw=gu.dev(dw0.clone()), db=gu.dev(db0.clone()), alpha=alpha, swap_hw=swap)
    for p in range(P):
        wr = torch.zeros(16, 16, 5, 5, requires_grad=True)
        br = torch.zeros(16, requires_grad=True)
        F.conv2d(x[p], wr, br, padding=2).backward(gy[p])
        gw = wr.grad.transpose(2, 3) if swap else wr.grad
        scale = max(float(gw.abs().max()), 1e-12)
        assert maxdiff(dw[p].cpu() - dw0[p], alpha * gw) < 2e-5 * abs(alpha) * scale + 5e-7, (geom, p)      # + the rounding of dw0 + small
        assert maxdiff(db[p].cpu() - db0[p], alpha * br.grad) < 2e-5 * abs(alpha) * float(br.grad.abs().max()) + 2e-7, (geom, p)


@pytest.mark.parametrize("dims,hw,PB", [((162, 162, 54, 18), (9, 13), (2, 2)), ((20, 12, 8, 4), (8, 24), (2, 2)),
                                        # 24 576 pixels = 384 columns of 64: every persistent workgroup
                                        # of k_cgp_rate<true> / k_cgp_bwd walks several 8-column groups with register prefetch
                                        ((162, 162, 54, 18), (128, 192), (1, 1)), ((162, 162, 54, 18), (67, 131), (2, 1))])
def test_cgp_fused_forward_backward(dims, hw, PB):
    """CgpRateFn (fused cgp stack + Gaussian rate, forward and backward) vs torch autograd over the same maths: ragged
    pixel count (not a multiple of the 64-pixel column), the reference widths and a narrow stack, and sizes at which the
    kernels' multi-column persistence runs (LiftingBasedDWT_net.py:282-289,357-365)."""
    ag, ops, gu = _mods()
    g = torch.Generator().manual_seed(sum(dims))
    P, B = PB
    G = 3
    h, w = hw
    c = list(dims) + [2]
    cat = torch.randn(P, B, G * c[0], h, w, generator=g)
    x = torch.randn(P, B, G, h, w, generator=g) * 2
    noise = torch.rand(P, B, G, h, w, generator=g) - 0.5
    ws = [torch.randn(P, G * c[l + 1], c[l], 1, 1, generator=g) * (1.5 / c[l] ** 0.5) for l in range(4)]
    bs = [torch.randn(P, G * c[l + 1], generator=g) * 0.1 for l in range(4)]
    bs[3] = bs[3] + torch.tensor([1.0, 0.0] * G)                      # keep sigma away from the 0.11 clamp mostly
    gb = torch.rand(P, B, G, h, w, generator=g)
    dv = [gu.dev(t).requires_grad_(True) for t in [cat, x] + [t for pair in zip(ws, bs) for t in pair]]
    bits = ag.CgpRateFn.apply(dv[0], dv[1], gu.dev(noise), G, *dv[2:])
    bits.backward(gu.dev(gb))
    for p in range(P):
        rv = [t[p].clone().requires_grad_(True) for t in [cat, x] + [t for pair in zip(ws, bs) for t in pair]]
        t = rv[0]
        for l in range(4):
            t = F.conv2d(t, rv[2 + 2 * l], rv[3 + 2 * l], groups=G)
            if l < 3:
                t = F.leaky_relu(t, 0.01)
        sigma, mu = t[:, 0::2], t[:, 1::2]
        lik = entropy.gaussian_likelihood(rv[1] + noise[p], sigma, mu)
        ref = -torch.log2(lik)
        assert maxdiff(bits[p].detach().cpu(), ref) < 2e-4
        ref.backward(gb[p])
        for a, b, name in zip(dv, rv, ["cat", "x"] + ["w%d" % (i // 2) if i % 2 == 0 else "b%d" % (i // 2) for i in range(8)]):
            scale = max(1.0, float(b.grad.abs().max()))
            assert maxdiff(a.grad[p].cpu(), b.grad) < 5e-4 * scale, name


@pytest.mark.parametrize("fwd,bwd", [("f32", "f32"), ("f16x3", "f32"), ("f32", "f16x3"), ("f16x3", "f16x3")])
@pytest.mark.parametrize("hw,PB", [((9, 13), (2, 2)), ((67, 131), (2, 1)), ((128, 192), (1, 2))])
def test_cgp_ctx_training_path_equals_the_concatenated_one(hw, PB, fwd, bwd, monkeypatch):
    """CgpRateCtxFn (lldwt_cgp_rate_train_ctx / lldwt_cgp_bwd_split / lldwt_wgrad1x1_split: the kernels read the tree-context
    tensor and gather the causal taps of the quantised subband themselves) against CgpRateFn on the explicit [plc_g | taps_g]
    concatenation with autograd through the gather -- bits, and the gradients of plc, the quantised subband, the coefficients and
    all eight parameter tensors (LiftingBasedDWT_net.py:282-289,353-365; the 5x5 type-A mask keeps 12 causal taps)."""
    ag, ops, gu = _mods()
    # fwd: CgpRateCtxFn's forward on the fp32-MFMA kernel (the same kernel as CgpRateFn's: bit-identical bits) or on the split-fp16
    # register chain with the hidden activations written out (lldwt_cgp16_params_train + lldwt_gauss_rate, the default): fp32-level
    # accuracy, so bits within 2e-4 and gradients within 2e-4 of each tensor's maximum
    # bwd: the backward-data chain on the fp32-MFMA kernel (lldwt_cgp_bwd_split) or on the split-fp16 register chain (lldwt_cgp16_bwd)
    monkeypatch.setattr(ag, "_CGP_TRAIN_F16", fwd == "f16x3")
    monkeypatch.setattr(ag, "_CGP_TRAIN_BWD_F16", bwd == "f16x3")
    g = torch.Generator().manual_seed(hw[0])
    P, B = PB
    G, K, cplc = 3, 5, 81
    live = list(range(12))                                             # rows 0, 1 and the first two taps of row 2
    tap_mask = sum(1 << t for t in live)
    h, w = hw
    c = [cplc + len(live), 162, 54, 18, 2]
    plc = torch.randn(P, B, G * cplc, h, w, generator=g)
    xq = torch.round(torch.randn(P, B, G, h, w, generator=g) * 3) + (torch.rand(P, B, G, h, w, generator=g) - 0.5)
    x = torch.randn(P, B, G, h, w, generator=g) * 2
    noise = torch.rand(P, B, G, h, w, generator=g) - 0.5
    ws = [torch.randn(P, G * c[l + 1], c[l], 1, 1, generator=g) * (1.5 / c[l] ** 0.5) for l in range(4)]
    bs = [torch.randn(P, G * c[l + 1], generator=g) * 0.1 for l in range(4)]
    bs[3] = bs[3] + torch.tensor([1.0, 0.0] * G)
    gb = torch.rand(P, B, G, h, w, generator=g)
    flat = [plc, xq, x] + [t for pair in zip(ws, bs) for t in pair]

    def run(ctx):
        dv = [gu.dev(t).requires_grad_(True) for t in flat]
        if ctx:
            bits = ag.CgpRateCtxFn.apply(dv[0], dv[1], dv[2], gu.dev(noise), G, K, tap_mask, *dv[3:])
        else:
            patches = ag.CgpRateCtxFn._gather(dv[1], K, live)          # differentiable torch ops, as _fold_csc_train builds them
            pl, pa = dv[0].chunk(G, dim=2), patches.chunk(G, dim=2)
            cat = torch.cat([z for gi in range(G) for z in (pl[gi], pa[gi])], dim=2).contiguous()
            bits = ag.CgpRateFn.apply(cat, dv[2], gu.dev(noise), G, *dv[3:])
        bits.backward(gu.dev(gb))
        return bits.detach(), [t.grad for t in dv]
    b1, g1 = run(True)
    b0, g0 = run(False)
    if fwd == "f32":
        assert torch.equal(b1, b0)                                      # the same kernel on the same values
    else:
        # random weights put coefficients far into the tails (bits up to ~30, where d bits / d mu is ~100): the 2^-22 operand
        # rounding of the split arithmetic shows as up to ~1e-3 there; the bars at realistic weights are the oracle tests'
        assert float((b1 - b0).abs().mean()) < 2e-5 and float((b1 - b0).abs().max()) < 3e-3
    names = ["plc", "xq", "x"] + ["w%d" % (i // 2) if i % 2 == 0 else "b%d" % (i // 2) for i in range(8)]
    tol = 2e-5 if (fwd, bwd) == ("f32", "f32") else 5e-4
    for a, b, n in zip(g1, g0, names):
        assert float((a - b).abs().max()) <= tol * max(1e-6, float(b.abs().max())), n


@pytest.mark.parametrize("fused_wgrad", [True, False])
@pytest.mark.parametrize("C,hw", [(3, (7, 19)), (1, (16, 40)), (3, (72, 100))])
def test_subband_mlp_fused_forward_backward(C, hw, fused_wgrad, monkeypatch):
    """SubbandMlpFn (in-register MFMA MLP, forward and backward) vs torch autograd: ragged coefficient counts (not multiples of
    the 32/64-coefficient wave tiles), 1 and 3 channels, one plane large enough for several rounds of every wave.  fused_wgrad:
    the eight parameter gradients formed inside the backward kernel (lldwt_subband_mlp_bwd_w, the default) or by the four
    grouped 1x1 weight-gradient GEMMs over tensors written out (LLDWT_MLP_WGRAD=gemm)."""
    ag, ops, gu = _mods()
    monkeypatch.setattr(ag, "_MLP_WGRAD_FUSED", fused_wgrad)
    g = torch.Generator().manual_seed(40 + C)
    P, B, H = 2, 2, 32
    h, w = hw
    x = torch.randn(P, B, C, h, w, generator=g)
    shapes = [(C * H, 1), (C * H, H), (C * H, H), (C, H)]
    ws = [torch.randn(P, o, i, 1, 1, generator=g) * (1.2 / i ** 0.5) for o, i in shapes]
    bs = [torch.randn(P, o, generator=g) * 0.2 for o, _ in shapes]
    gy = torch.randn(P, B, C, h, w, generator=g)
    dv = [gu.dev(t).requires_grad_(True) for t in [x] + [t for pair in zip(ws, bs) for t in pair]]
    y = ag.SubbandMlpFn.apply(*dv)
    y.backward(gu.dev(gy))
    for p in range(P):
        rv = [t[p].clone().requires_grad_(True) for t in [x] + [t for pair in zip(ws, bs) for t in pair]]
        t = rv[0]
        for l in range(4):
            t = F.conv2d(t, rv[1 + 2 * l], rv[2 + 2 * l], groups=C)
            if l < 3:
                t = torch.tanh(t)
        assert maxdiff(y[p].detach().cpu(), t) < 2e-5
        t.backward(gy[p])
        for i, (a, b) in enumerate(zip(dv, rv)):
            scale = max(1.0, float(b.grad.abs().max()))
            assert maxdiff(a.grad[p].cpu(), b.grad) < 2e-4 * scale, i


def _lift_stacks(sds, nblocks, gu):
    """oracle per-plane state dicts -> taps (4,P,3) and the 8 stacked tensors (nblocks,2,P,...) on the device."""
    taps = torch.stack([torch.stack([sd["preProcessingList.%d.weight" % j].reshape(3) for sd in sds], 0) for j in range(4)], 0)
    W = []
    for n in (1, 2, 3, 4):
        for k in ("weight", "bias"):
            W.append(torch.stack([torch.stack([torch.stack([sd["%s.%d.conv%d.%s" % (kind, b, n, k)] for sd in sds], 0)
                                               for kind in ("P_blocks", "U_blocks")], 0) for b in range(nblocks)], 0))
    return gu.dev(taps), [gu.dev(t) for t in W]


@pytest.mark.parametrize("K,different,linear,scale,geom", [
    (5, False, False, False, (2, 2, 16, 32)), (3, True, False, False, (2, 2, 16, 32)), (3, False, True, False, (2, 2, 16, 32)),
    (5, False, False, True, (2, 2, 16, 32)), (3, False, True, True, (2, 2, 16, 32)),
    # sizes where the 16x32 lifting tiles have INTERIOR tiles and a persistent workgroup walks several
    # tiles (level-0 half arrays 128x96 / 64x96 ..., level 1 64x48): the fast paths the training leg of bench.py runs
    (5, False, False, False, (1, 1, 128, 192)), (3, True, False, True, (1, 2, 96, 160))])
def test_lifting_forward_backward(K, different, linear, scale, geom):
    """LiftingFn / LiftingInvFn gradients (input, skip-filter taps, every P/U-block parameter, and with config.scale == 1 the
    gains nh / nl of wavelet_forward_v2.py:76-80) vs torch autograd on the oracle -- at border-only sizes and at sizes with
    interior tiles (agents/liftingDWT_agent.py:96-98 through wavelet_forward_v2.py:58-81)."""
    ag, ops, gu = _mods()
    from helpers import filled
    from oracle import lifting, model, weights
    L = 2
    P, B, H, W = geom
    cfg = dict(model.DEFAULT_CFG, dwtlevels=L, filtersize=K, block_property="different" if different else "same",
               linearity_flag=0 if linear else 1, scale=1 if scale else 0)
    nblocks = 2 * 2 * L if different else 2
    sds = [filled(weights.autoencoder_template(cfg), "bw%d." % p) for p in range(P)]

    def gains():
        if not scale:
            return None, None
        nh = torch.stack([lifting.LIFTING_COEFF[4] + sd["nh"].reshape(()) * 0.1 for sd in sds]).float()
        nl = torch.stack([lifting.LIFTING_COEFF[5] + sd["nl"].reshape(()) * 0.1 for sd in sds]).float()
        return gu.dev(nh).requires_grad_(True), gu.dev(nl).requires_grad_(True)
    nh, nl = gains()
    nh2, nl2 = gains()
    meta = dict(levels=L, C=16, K=K, rw=0.1, linear=linear, different=different)
    g = torch.Generator().manual_seed(21)
    x = torch.rand(P, B, 1, H, W, generator=g) - 0.5
    taps, Wt = _lift_stacks(sds, nblocks, gu)
    taps.requires_grad_(True)
    for t in Wt:
        t.requires_grad_(True)
    xd = gu.dev(x).requires_grad_(True)
    outs = ag.LiftingFn.apply(xd, taps, meta, nh, nl, *Wt)
    gouts = [torch.rand(o.shape, generator=g) - 0.5 for o in outs]
    torch.autograd.backward(outs, [gu.dev(t) for t in gouts])
    # inverse on the same coefficients
    taps2, Wt2 = _lift_stacks(sds, nblocks, gu)
    taps2.requires_grad_(True)
    for t in Wt2:
        t.requires_grad_(True)
    cin = [o.detach().clone().requires_grad_(True) for o in outs]
    xr = ag.LiftingInvFn.apply(taps2, meta, L, nh2, nl2, *cin, *Wt2)
    gx = torch.rand(xr.shape, generator=g) - 0.5
    xr.backward(gu.dev(gx))
    for p in range(P):
        sd = {k: v.clone().requires_grad_(True) for k, v in sds[p].items()}
        xp = x[p].clone().requires_grad_(True)
        oLL, oYh = lifting.lifting_forward(xp, sd, cfg)
        ref = [oLL] + [t[:, 0] for t in oYh]
        for a, b_ in zip(outs, ref):
            assert maxdiff(a[p].detach().cpu(), b_) < 1e-4
        torch.autograd.backward(ref, [t[p] for t in gouts])
        assert maxdiff(xd.grad[p].cpu(), xp.grad) < 2e-4
        for j in range(4):
            rt = sd["preProcessingList.%d.weight" % j].grad.reshape(3)        # sums over every pixel: relative at large sizes
            assert maxdiff(taps.grad[j, p].cpu(), rt) < 2e-3 * max(1.0, float(rt.abs().max())), j
        idx = 0
        for n in (1, 2, 3, 4):
            for k in ("weight", "bias"):
                for b in range(nblocks):
                    for u, kind in enumerate(("P_blocks", "U_blocks")):
                        r = sd["%s.%d.conv%d.%s" % (kind, b, n, k)].grad
                        got = Wt[idx].grad[b, u, p].cpu()
                        if r is None:       # block not used by the forward of this configuration
                            assert float(got.abs().max()) == 0.0
                        else:
                            assert maxdiff(got, r) < 5e-4 * max(1.0, float(r.abs().max())), (n, k, b, kind)
                idx += 1
        if scale:       # d/d(sd.nh) = 0.1 d/d(gain)
            for got, key in ((nh, "nh"), (nl, "nl")):
                r = sd[key].grad.reshape(())
                assert abs(float(got.grad[p]) * 0.1 - float(r)) < 5e-4 * max(1.0, abs(float(r))), key
        # inverse
        sd2 = {k: v.clone().requires_grad_(True) for k, v in sds[p].items()}
        ci = [c[p].detach().cpu().clone().requires_grad_(True) for c in cin]
        oxr = lifting.lifting_inverse(ci[0], [t.unsqueeze(1) for t in ci[1:]], sd2, cfg)
        assert maxdiff(xr[p].detach().cpu(), oxr) < 1e-4
        oxr.backward(gx[p])
        for a, b_ in zip(cin, ci):
            assert maxdiff(a.grad[p].cpu(), b_.grad) < 2e-4
        for j in range(4):
            rt = sd2["preProcessingList.%d.weight" % j].grad.reshape(3)
            assert maxdiff(taps2.grad[j, p].cpu(), rt) < 2e-3 * max(1.0, float(rt.abs().max())), j
        r = sd2["U_blocks.%d.conv2.weight" % (2 * L if different else 0)].grad
        assert maxdiff(Wt2[2].grad[2 * L if different else 0, 1, p].cpu(), r) < 5e-4 * max(1.0, float(r.abs().max()))
        if scale:
            for got, key in ((nh2, "nh"), (nl2, "nl")):
                r = sd2[key].grad.reshape(())
                assert abs(float(got.grad[p]) * 0.1 - float(r)) < 5e-4 * max(1.0, abs(float(r))), key


def test_gauss_rate_backward():
    ag, ops, gu = _mods()
    g = torch.Generator().manual_seed(31)
    P, B, C, h, w = 2, 2, 3, 6, 10
    x = (torch.rand(P, B, C, h, w, generator=g) - 0.5) * 10
    params = torch.rand(P, B, 2 * C, h, w, generator=g) * 3 - 0.4          # includes sigma < 0.11 (LowerBound rule)
    noise = torch.rand(P, B, C, h, w, generator=g) - 0.5
    gb = torch.rand(P, B, C, h, w, generator=g) - 0.3
    xd, pd = gu.dev(x).requires_grad_(True), gu.dev(params).requires_grad_(True)
    bits = ag.GaussRateFn.apply(xd, pd, gu.dev(noise))
    bits.backward(gu.dev(gb))
    xr, pr = x.clone().requires_grad_(True), params.clone().requires_grad_(True)
    for p in range(P):
        _, lik = entropy.gaussian_conditional_forward(xr[p], pr[p][:, 0::2], pr[p][:, 1::2], True, noise[p])
        (-torch.log2(lik) * gb[p]).sum().backward()
    assert maxdiff(xd.grad.cpu(), xr.grad) < 2e-4 * max(1.0, float(xr.grad.abs().max()))
    assert maxdiff(pd.grad.cpu(), pr.grad) < 2e-4 * max(1.0, float(pr.grad.abs().max()))


def test_colour_and_loss_backward():
    ag, ops, gu = _mods()
    from oracle import model
    g = torch.Generator().manual_seed(41)
    B, H, W = 2, 8, 12
    ycc = torch.rand(3, B, 1, H, W, generator=g) - 0.5
    tgt = torch.rand(B, 3, H, W, generator=g) - 0.5
    yd = gu.dev(ycc).requires_grad_(True)
    rgb = ag.YccToRgbFn.apply(yd)
    se = ag.SqErrSumFn.apply(gu.dev(tgt), rgb)
    (se / rgb.numel()).sum().backward()
    yr = ycc.clone().requires_grad_(True)
    rr = model.ycbcr2rgb(yr[:, :, 0].permute(1, 0, 2, 3) + model._YSHIFT) - 0.5
    torch.mean((tgt - rr) ** 2).backward()
    assert maxdiff(yd.grad.cpu(), yr.grad) < 1e-6
    t = gu.dev(torch.rand(3, 4, 5, generator=g)).requires_grad_(True)
    (ag.SumFn.apply(t) * 0.25).sum().backward()
    assert maxdiff(t.grad.cpu(), torch.full((3, 4, 5), 0.25)) < 1e-7


@pytest.mark.parametrize("hw", [(40, 72), (19, 33), (96, 160), (512, 512), (520, 64)])
@pytest.mark.parametrize("vertical", [True, False])
def test_lift_step_backward_on_the_fused_kernel(hw, vertical):
    """lldwt_lift_step_bwd_f16 (backward-data chain as one launch of the fused split-fp16 kernel, transposed + mirrored weights
    from lldwt_pack_pblock_bwd) against lldwt_lift_step_bwd (three fp32-MFMA launches) on the same saved intermediates: the
    chain's four gradients, the step's input gradient, tap and weight gradients.  Split-fp16 class: 2e-5 of each tensor's
    maximum (the lifting-forward tests compare both classes with the CPU restatement; P_block_v2.py:40-55 is what they
    differentiate).  At 512 x 512 (>= 0.5 Mpixel per plane) the 16 -> 16 weight gradients run on the split-fp16 kernel, and on
    the fused side take their dY scale from the maxima the backward launch leaves in the workspace slots (no pass over dY)."""
    import ctypes
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    torch.manual_seed(5)
    dev = torch.device("cuda:0")
    P, B, C, K = 3, 2, 16, 5
    h, w = hw
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
    saved[2 * n:2 * n + 2 * n * C] = torch.tanh(2 * torch.randn(2 * n * C, device=dev))        # t1, t2 (some saturated)
    saved[2 * n + 2 * n * C:] = torch.randn(n * C, device=dev)
    gout = torch.randn(Z, h, w, device=dev) * 3.0
    # (520, 64): enough tiles for runs of two (the launch picks the run length), and the gradient grows by six decades down the image -- the tiles of a run then differ in their bound-based operand
    # scales, so the T2 rows a tile hands down are RESCALED by the tile below (a height that is not a multiple of the tile, too)
    ramp = (10.0 ** torch.linspace(-3, 3, h, device=dev))[None, :, None] if hw == (520, 64) else None
    if ramp is not None:
        gout = gout * ramp
    if not ops.bwd_lift_f16():
        pytest.skip("LLDWT_BWD_LIFT=f32 / lift mode f32: the fused backward is switched off")
    res = []
    for fused in (False, True):
        gdin, gsrc = torch.zeros(Z, h, w, device=dev), torch.zeros(Z, h, w, device=dev)
        dW = [torch.zeros_like(W[k]) for k in keys]
        dtaps = torch.zeros_like(taps)
        v = lambda t: ops.View(ctypes.c_void_p(t.data_ptr()), h * w, w, 1)
        ops.lift_step_bwd(v(gout), v(gdin), v(gsrc), saved, P, B, h, w, taps, dtaps, ctypes.c_void_p(packed.data_ptr()),
                          packed.shape[1], dW, C, K, 0.5, -1.0, vertical, False,
                          packed_bwd=ctypes.c_void_p(bpack.data_ptr()) if fused else None, taps_id=tid if fused else None)
        torch.cuda.synchronize()
        ws = ops.workspace(0, dev).view(torch.float32)
        chain = {"dsk": ws[n:2 * n], "dt3": ws[2 * n:2 * n + n * C], "dpre2": ws[2 * n + n * C:2 * n + 2 * n * C],
                 "dr": ws[2 * n + 2 * n * C:2 * n + 3 * n * C]}
        out = {k: t.clone() for k, t in chain.items()}
        out.update(gdin=gdin, gsrc=gsrc, dtaps=dtaps)
        out.update({"d" + k: t for k, t in zip(keys, dW)})
        res.append(out)
    ref, got = res
    assert torch.equal(ref["gdin"], got["gdin"])
    for k in ref:
        scale = float(ref[k].abs().max())
        assert scale > 0, k
        assert float((ref[k] - got[k]).abs().max()) <= 2e-5 * scale, (k, float((ref[k] - got[k]).abs().max()), scale)
    if ramp is not None:        # per-pixel tensors scale with the gradient: compare them row by row, not against the global maximum
        for k in ("dsk", "dt3", "dpre2", "dr", "gsrc"):
            r_, g_ = ref[k].view(Z, -1, h, w) / ramp[:, None], got[k].view(Z, -1, h, w) / ramp[:, None]
            assert float((r_ - g_).abs().max()) <= 4e-5 * float(r_.abs().max()), k


def test_new_backward_entries_refuse_bad_arguments():
    """The round-3 backward entry points fail loudly (LLDWT_EINVAL / LLDWT_EWS -> LLDWTError), never silently: a backward pack for
    a block the fused kernel is not built for, a fused step backward without its pack, a workspace that is too small."""
    import ctypes
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import _lib, ops
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd._lib import LLDWTError
    dev = torch.device("cuda:0")
    lib = _lib.load()
    P = 2
    w1, w2 = torch.zeros(P, 16, 1, 3, 3, device=dev), torch.zeros(P, 16, 16, 3, 3, device=dev)
    w4 = torch.zeros(P, 1, 16, 3, 3, device=dev)
    with pytest.raises(LLDWTError, match="K=5"):
        ops.pack_pblock_bwd(w1, w2, w2, w4)                                    # 3 x 3 block: the fused kernel is 5 x 5 only
    x = torch.zeros(P, 2, 3, 8, 8, device=dev)
    par = [torch.zeros(P, 96, 1, 1, 1, device=dev), torch.zeros(P, 96, device=dev), torch.zeros(P, 96, 32, 1, 1, device=dev),
           torch.zeros(P, 96, device=dev), torch.zeros(P, 96, 32, 1, 1, device=dev), torch.zeros(P, 96, device=dev),
           torch.zeros(P, 3, 32, 1, 1, device=dev)]
    grads = [torch.zeros(4096, device=dev) for _ in range(8)]
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    ws = torch.zeros(64, device=dev)
    rc = lib.lldwt_subband_mlp_bwd_w(ptr(x), ptr(x), ptr(torch.empty_like(x)), P, 2, 3, 64, 32, *[ptr(t) for t in par],
                                     *[ptr(t) for t in grads], ptr(ws), 256, None)
    assert rc == -3 and b"workspace" in lib.lldwt_last_error()                  # LLDWT_EWS
    rc = lib.lldwt_subband_mlp_bwd_w(ptr(x), ptr(x), ptr(torch.empty_like(x)), P, 2, 3, 64, 16, *[ptr(t) for t in par],
                                     *[ptr(t) for t in grads], ptr(ws), 256, None)
    assert rc == -1 and b"hidden width" in lib.lldwt_last_error()               # LLDWT_EINVAL: H = 32 only
    v = ops.View(ptr(x), 64, 8, 1)
    with pytest.raises(LLDWTError, match="null"):
        lib_rc = lib.lldwt_lift_step_bwd_f16(v, v, v, ptr(x), P, 2, 8, 8, ptr(x), ptr(x), ptr(x), 0, *[ptr(x)] * 8, 16, 5, 0.1, 1.0,
                                             1, 0, ptr(x), 1 << 20, None, None, None)
        _lib.check(lib_rc, "lift_step_bwd_f16")


def test_late_round3_entries_refuse_bad_arguments():
    """The entry points added at the end of round 3 fail loudly (LLDWTError), never silently: the register-chain packs for other
    widths than the reference's, the split weight gradient for a tile plan it is not built for, a tree-conv weight gradient whose
    |max| slots have the wrong size, null pointers."""
    import ctypes
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import _lib, ops
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd._lib import LLDWTError
    dev = torch.device("cuda:0")
    lib = _lib.load()
    P, G = 1, 3
    ws = [torch.zeros(P, G * co, ci, 1, 1, device=dev) for co, ci in ((20, 12), (8, 20), (4, 8), (2, 4))]
    with pytest.raises(LLDWTError, match="93"):
        ops.cgp16_pack_bwd(ws, G)                                               # the chain is built for 93 -> 162 -> 54 -> 18 -> 2 only
    assert lib.lldwt_cgp16_bwd_packed_bytes(12, 20, 8, 4, G) == -1
    xa, xb = torch.zeros(P, 1, G * 8, 4, 4, device=dev), torch.zeros(P, 1, G * 4, 4, 4, device=dev)
    with pytest.raises(LLDWTError, match="built for"):
        ops.wgrad1x1_split(xa, xb, torch.zeros(P, 1, G * 20, 4, 4, device=dev), G)   # 12 input rows: below the tile plan
    x = torch.zeros(P, 1, 64, 8, 8, device=dev)
    with pytest.raises(LLDWTError, match="slots"):
        ops.conv3x3_wgrad_f16x3(x, x, (P, 64, 64, 3, 3), x_slots=torch.zeros(P, 32, device=dev))
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.lldwt_cgp16_bwd(ptr(x), ptr(x), ptr(x), ptr(x), None, ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), 1, 1, 64, G, None)
    assert rc == -1 and b"cgp16_bwd" in lib.lldwt_last_error()                  # LLDWT_EINVAL: null pack
    rc = lib.lldwt_cgp_rate_train_ctx(ptr(x), None, ptr(x), None, ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), 1, 1, 8, 8, 81, 5,
                                      0xFFF, 162, 54, 18, G, None)
    assert rc == -1 and b"cgp_rate_train_ctx" in lib.lldwt_last_error()
