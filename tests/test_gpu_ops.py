"""GPU parity: the small kernels (colour, subband MLP, conv, GDN, bound ops, rate estimation, CDF 9/7) vs the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import GOLDEN, filled, load_golden, maxdiff
from oracle import cdf97, entropy, model, subband_ae, weights

pytestmark = pytest.mark.gpu


def _ops():
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    import gpu_util
    return ops, gpu_util


def test_colour_round_trip_and_oracle():
    ops, gu = _ops()
    x = torch.rand(3, 3, 17, 33, generator=torch.Generator().manual_seed(1))
    y = ops.rgb_to_ycc(gu.dev(x))
    ref = model.rgb2ycbcr(x) - model._YSHIFT
    assert maxdiff(y[:, :, 0].permute(1, 0, 2, 3).cpu(), ref) < 1e-6
    back = ops.ycc_to_rgb(y)
    assert maxdiff(back.cpu(), x - 0.5) < 1e-6
    y2 = y.clone()
    y2[0] += 3.0
    assert float(ops.ycc_to_rgb(y2, clamp=True).abs().max()) <= 0.5


@pytest.mark.parametrize("C", [1, 3])
def test_subband_mlp(C):
    ops, gu = _ops()
    cfg = dict(model.DEFAULT_CFG, dwtlevels=1)
    P, B, h, w = 2, 2, 9, 21
    sds = [filled(weights.autoencoder_template(cfg), "q%d." % p) for p in range(P)]
    pre = "Yl_ae." if C == 1 else "Yh_ae.0."
    x = (torch.rand(P, B, C, h, w, generator=torch.Generator().manual_seed(2)) - 0.5) * 4
    for updown, transposed, fn in (("ae_down", False, subband_ae.subband_ae_encode),
                                   ("ae_up", True, subband_ae.subband_ae_decode)):
        ws = [gu.stack(sds, pre + "%s.%d.%s" % (updown, n, k)).flatten(1) for n in (0, 2, 4, 6) for k in ("weight", "bias")]
        y = ops.subband_mlp(gu.dev(x), *ws, transposed=transposed)
        for p in range(P):
            assert maxdiff(y[p].cpu(), fn(x[p], sds[p], pre)) < 2e-5


CONV_CASES = [
    # cin, cout, K, groups, act, upsample, transposed, masktype
    (3, 243, 3, 1, 2, True, False, None),      # plc first layer on the 2x-upsampled parent
    (48, 40, 3, 1, 0, False, False, None),     # dense 3x3
    (3, 243, 5, 3, 0, False, False, "A"),      # csc masked 5x5 grouped
    (243, 81, 3, 3, 2, False, False, "B"),     # masked 3x3 B grouped
    (486, 162, 1, 3, 2, False, False, None),   # cgp 1x1 grouped
    (18, 6, 1, 3, 0, False, False, None),
    (12, 20, 3, 1, 0, False, True, None),      # ConvTranspose2d (Berk decoder)
    (16, 16, 5, 1, 1, False, False, None),     # tanh (lifting P-block conv2)
    (243, 243, 3, 1, 0, False, False, None),   # plc second conv: the dominant layer
    (3, 6, 3, 3, 0, False, False, "A"),        # tiny grouped masked (cout/groups = 2)
    (27, 9, 3, 1, 2, False, False, "B"),       # csc_xe tail
    (96, 192, 3, 1, 0, False, False, None),    # Berk auto-encoder middle layer
    (1, 32, 1, 1, 1, False, False, None),      # cin = 1, 1x1
]


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case, direct):
    """direct=False: the MFMA implicit-GEMM engine; direct=True: the reference-order VALU kernel."""
    ops, gu = _ops()
    cin, cout, K, groups, act, up, tr, mt = case
    P, B, h, w = 2, 2, 10, 14
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.rand(P, B, cin, h, w, generator=g) - 0.5
    if tr:
        wt = (torch.rand(P, cin, cout, K, K, generator=g) - 0.5) * 0.3
    else:
        wt = (torch.rand(P, cout, cin // groups, K, K, generator=g) - 0.5) * 0.3
    bias = torch.rand(P, cout, generator=g) - 0.5
    mask_bits = None
    if mt:
        m = entropy.conv_mask((cout, cin // groups, K, K), mt)
        wt = wt * m
        mask_bits = int(sum(1 << t for t in range(K * K) if m[0, 0].flatten()[t] > 0))
    y = ops.conv2d(gu.dev(x), gu.dev(wt), gu.dev(bias), K, groups=groups, act=act, upsample2=up, transposed=tr,
                   tap_mask=mask_bits, direct=direct)
    for p in range(P):
        xi = entropy.upsample2(x[p]) if up else x[p]
        if tr:
            ref = F.conv_transpose2d(xi, wt[p], bias[p], padding=K // 2)
        else:
            ref = F.conv2d(xi, wt[p], bias[p], padding=K // 2, groups=groups)
        ref = torch.tanh(ref) if act == 1 else (F.leaky_relu(ref, 0.01) if act == 2 else ref)
        assert maxdiff(y[p].cpu(), ref) < 2e-5, case


def test_conv2d_channel_placement():
    """plc/csc outputs written straight into the interleaved (plc_g, csc_g) layout of LiftingBasedDWT_net.py:357-359."""
    ops, gu = _ops()
    P, B, h, w = 1, 1, 6, 7
    g = torch.Generator().manual_seed(4)
    xa = torch.rand(P, B, 3, h, w, generator=g)
    wa = torch.rand(P, 243, 3, 3, 3, generator=g) - 0.5
    wb = (torch.rand(P, 243, 1, 5, 5, generator=g) - 0.5) * entropy.conv_mask((243, 1, 5, 5), "A")
    out = torch.zeros(P, B, 486, h, w, device=gu.DEV)
    ops.conv2d(gu.dev(xa), gu.dev(wa), None, 3, out=out, oc_block=81, oc_stride=162, oc_off=0)
    ops.conv2d(gu.dev(xa), gu.dev(wb), None, 5, groups=3, out=out, oc_block=81, oc_stride=162, oc_off=81,
               tap_mask=sum(1 << t for t in range(12)))
    plc = F.conv2d(xa[0], wa[0], None, padding=1)
    csc = F.conv2d(xa[0], wb[0], None, padding=2, groups=3)
    p0, p1, p2 = plc.chunk(3, 1)
    c0, c1, c2 = csc.chunk(3, 1)
    ref = torch.cat((p0, c0, p1, c1, p2, c2), 1)
    assert maxdiff(out[0].cpu(), ref) < 2e-5


def test_conv2d_mfma_ragged_and_residual():
    """Image sizes that are not multiples of any tile, plus the residual epilogue (P_block_v2.py:53)."""
    ops, gu = _ops()
    g = torch.Generator().manual_seed(11)
    for (cin, cout, K, h, w) in [(16, 16, 5, 37, 53), (243, 243, 3, 9, 19), (20, 100, 1, 5, 70)]:
        x = torch.rand(1, 2, cin, h, w, generator=g) - 0.5
        wt = (torch.rand(1, cout, cin, K, K, generator=g) - 0.5) * 0.2
        b = torch.rand(1, cout, generator=g) - 0.5
        res = torch.rand(1, 2, cout, h, w, generator=g) - 0.5
        y = ops.conv2d(gu.dev(x), gu.dev(wt), gu.dev(b), K, act=1, residual=gu.dev(res))
        ref = torch.tanh(F.conv2d(x[0], wt[0], b[0], padding=K // 2) + res[0])
        assert maxdiff(y[0].cpu(), ref) < 2e-5, (cin, cout, K)
        yd = ops.conv2d(gu.dev(x), gu.dev(wt), gu.dev(b), K, direct=True)
        ym = ops.conv2d(gu.dev(x), gu.dev(wt), gu.dev(b), K)
        assert maxdiff(yd, ym) < 2e-5


def test_gdn_and_bound_ops():
    ops, gu = _ops()
    g = load_golden("ref_gdn")
    tpl = {"Yl_ae.ae_down.1.beta": subband_ae.nonneg_init(torch.ones(6)),
           "Yl_ae.ae_down.1.gamma": subband_ae.nonneg_init(0.1 * torch.eye(6))}
    sd = filled(tpl)
    b, gm = sd["Yl_ae.ae_down.1.beta"], sd["Yl_ae.ae_down.1.gamma"]
    y = ops.gdn(gu.pm(g["x"]), gu.dev(b)[None], gu.dev(gm)[None], False)
    yi = ops.gdn(gu.pm(g["x"]), gu.dev(b)[None], gu.dev(gm)[None], True)
    assert maxdiff(y[0].cpu(), g["y"]) < 2e-6
    assert maxdiff(yi[0].cpu(), g["y_inv"]) < 2e-6
    lb = load_golden("ref_lower_bound")
    assert torch.equal(ops.lower_bound_fwd(gu.dev(lb["x"]), 0.11).cpu(), lb["y"])
    assert torch.equal(ops.lower_bound_bwd(gu.dev(lb["x"]), gu.dev(lb["gup"]), 0.11).cpu(), lb["gx"])
    nn_ = load_golden("ref_nonneg_param")
    assert maxdiff(ops.nonneg_param_fwd(gu.dev(nn_["x"]), 1e-6).cpu(), nn_["y"]) < 1e-12
    assert maxdiff(ops.nonneg_param_bwd(gu.dev(nn_["x"]), torch.ones(6, device=gu.DEV), 1e-6).cpu(), nn_["gx"]) < 1e-12


def test_gauss_rate_eval_and_noise():
    ops, gu = _ops()
    g = torch.Generator().manual_seed(6)
    P, B, C, h, w = 2, 2, 3, 8, 9
    x = (torch.rand(P, B, C, h, w, generator=g) - 0.5) * 12
    params = torch.rand(P, B, 2 * C, h, w, generator=g) * 3 - 0.5      # some sigmas below the 0.11 bound / negative
    bsum = torch.zeros(1, dtype=torch.float64, device=gu.DEV)
    bits, q = ops.gauss_rate(gu.dev(x), gu.dev(params), None, want_q=True, bit_sum=bsum)
    for p in range(P):
        sg, mu = params[p][:, 0::2], params[p][:, 1::2]
        oq, lik = entropy.gaussian_conditional_forward(x[p], sg, mu, False)
        assert maxdiff(q[p].cpu(), oq) < 1e-6
        assert maxdiff(bits[p].cpu(), -torch.log2(lik)) < 1e-4
    assert abs(float(bsum) - float(bits.double().sum())) < 1e-6 * float(bits.double().sum())
    noise = torch.rand(P, B, C, h, w, generator=g) - 0.5
    bits, q = ops.gauss_rate(gu.dev(x), gu.dev(params), gu.dev(noise), want_q=True)
    for p in range(P):
        sg, mu = params[p][:, 0::2], params[p][:, 1::2]
        oq, lik = entropy.gaussian_conditional_forward(x[p], sg, mu, True, noise[p])
        assert maxdiff(q[p].cpu(), oq) < 1e-6
        assert maxdiff(bits[p].cpu(), -torch.log2(lik)) < 1e-4
    assert torch.equal(ops.quantize(gu.dev(x)).cpu(), torch.round(x))


def test_cgp_fused_rate():
    """Fused cgp MLP + Gaussian rate vs the oracle's grouped 1x1 conv stack (LiftingBasedDWT_net.py:357-365)."""
    ops, gu = _ops()
    g = torch.Generator().manual_seed(12)
    P, B, G, h, w = 2, 2, 3, 9, 13          # 117 pixels: ragged 64-pixel columns
    dims = [162, 162, 54, 18, 2]
    cat = torch.rand(P, B, G * 162, h, w, generator=g) - 0.5
    x = (torch.rand(P, B, G, h, w, generator=g) - 0.5) * 8
    ws = [(torch.rand(P, G * dims[l + 1], dims[l], 1, 1, generator=g) - 0.5) * (3.0 / dims[l]) ** 0.5 * 2 for l in range(4)]
    bs = [(torch.rand(P, G * dims[l + 1], generator=g) - 0.5) * 0.4 for l in range(4)]
    packed, d = ops.cgp_pack([gu.dev(t) for t in ws], [gu.dev(t) for t in bs], G)
    bsum = torch.zeros(1, dtype=torch.float64, device=gu.DEV)
    bits, params = ops.cgp_rate(gu.dev(cat), gu.dev(x), packed, d, want_params=True, bit_sum=bsum)
    noise = torch.rand(P, B, G, h, w, generator=g) - 0.5
    bits_n, _ = ops.cgp_rate(gu.dev(cat), gu.dev(x), packed, d, noise=gu.dev(noise))
    for p in range(P):
        t = cat[p]
        for l in range(4):
            t = F.conv2d(t, ws[l][p], bs[l][p], groups=G)
            if l < 3:
                t = F.leaky_relu(t, 0.01)
        assert maxdiff(params[p].cpu(), t) < 2e-5
        _, lik = entropy.gaussian_conditional_forward(x[p], t[:, 0::2], t[:, 1::2], False)
        assert maxdiff(bits[p].cpu(), -torch.log2(lik)) < 2e-4
        _, lik = entropy.gaussian_conditional_forward(x[p], t[:, 0::2], t[:, 1::2], True, noise[p])
        assert maxdiff(bits_n[p].cpu(), -torch.log2(lik)) < 2e-4
    assert abs(float(bsum) - float(bits.double().sum())) < 1e-6 * float(bits.double().sum())


def test_factorized_rate():
    ops, gu = _ops()
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.entropy_models import pack_entropy_bottleneck
    P, B, C, h, w = 2, 2, 3, 7, 11
    sds = [weights.fill_by_name({"p%d.e." % p + a: b for a, b in entropy.eb_init_state(C).items()}) for p in range(P)]
    x = (torch.rand(P, B, C, h, w, generator=torch.Generator().manual_seed(8)) - 0.5) * 30
    eb = torch.stack([pack_entropy_bottleneck({k.split(".e.")[1]: v for k, v in sds[p].items()}) for p in range(P)], 0)
    bits, q = ops.factorized_rate(gu.dev(x), gu.dev(eb))
    for p in range(P):
        oq, lik = entropy.entropy_bottleneck_forward(x[p], sds[p], "p%d.e." % p, False)
        assert maxdiff(q[p].cpu(), oq) < 1e-6
        assert maxdiff(bits[p].cpu(), -torch.log2(lik)) < 2e-4
    # the call above read the per-offset table precomputed once per parameter version (lldwt_factorized_table +
    # lldwt_factorized_rate_tab); a parameter tensor that requires grad takes the kernel that builds the table per workgroup
    # (lldwt_factorized_rate): the same code evaluates the entries, the results are the same bits.  Values beyond the table
    # (|x - median| > 127) take the direct chain in both.
    xb = gu.dev(x).clone()
    xb[0, 0, 0, 0, :4] = torch.tensor([300.0, -250.5, 127.4, -128.6], device=xb.device)
    b_tab, q_tab = ops.factorized_rate(xb, gu.dev(eb))
    b_ker, q_ker = ops.factorized_rate(xb, gu.dev(eb).requires_grad_(True))
    assert torch.equal(b_tab, b_ker) and torch.equal(q_tab, q_ker)
    eb2 = gu.dev(eb)
    b0, _ = ops.factorized_rate(xb, eb2)
    eb2[:, :, 58] += 0.25                                   # in-place update of the medians: the cached table must not survive it
    b1, _ = ops.factorized_rate(xb, eb2)
    b1k, _ = ops.factorized_rate(xb, eb2.clone().requires_grad_(True))
    assert not torch.equal(b0, b1) and torch.equal(b1, b1k)


def test_cdf97_vs_pywt_and_round_trip():
    ops, gu = _ops()
    z = np.load(GOLDEN + "/cdf97_pywt.npz")
    x = torch.tensor(z["x"], dtype=torch.float32)          # (2,1,32,48)
    ll, yh = ops.cdf97_forward(gu.pm(x), 2)
    assert maxdiff(ll[0].cpu(), torch.tensor(z["ll"])) < 2e-5
    for i in range(2):
        for j, n in enumerate(("lh", "hl", "hh")):
            assert maxdiff(yh[i][0][:, :, j].cpu(), torch.tensor(z["%s%d" % (n, i)])) < 2e-5
    xr = ops.cdf97_inverse(ll, yh)
    assert maxdiff(xr[0].cpu(), x) < 2e-5
    # full-size round trip property (BASELINE config 1 size, 3 channels)
    big = torch.rand(1, 1, 3, 256, 256, device=gu.DEV)
    ll, yh = ops.cdf97_forward(big, 4)
    assert maxdiff(ops.cdf97_inverse(ll, yh), big) < 5e-5
    oll, oyh = cdf97.dwt_forward(big[0].cpu(), 4)
    assert maxdiff(ll[0].cpu(), oll) < 5e-5
    for i in range(4):
        assert maxdiff(yh[i][0].cpu(), oyh[i]) < 5e-5, i
    # ragged tiles of the fused level kernels (subband sizes that are not multiples of the 32x32 tile), non-square
    rag = torch.rand(1, 2, 1, 144, 200, device=gu.DEV) - 0.5
    ll, yh = ops.cdf97_forward(rag, 3)
    oll, oyh = cdf97.dwt_forward(rag[0].cpu(), 3)
    assert maxdiff(ll[0].cpu(), oll) < 5e-5
    for i in range(3):
        assert maxdiff(yh[i][0].cpu(), oyh[i]) < 5e-5, i
    assert maxdiff(ops.cdf97_inverse(ll, yh), rag) < 5e-5


def test_cdf97_short_levels_vs_pywt():
    """Level inputs shorter than the 10-tap filter (8, 6, 4, 2 samples): the kernels wrap every tap periodically, like
    PyWavelets (tests/golden/cdf97_pywt_small.npz) and the oracle's periodic=True form; the oracle's default form (single
    fold, as restated from pytorch_wavelets) differs there and only there -- see tests/test_oracle_golden.py and DESIGN.md."""
    ops, gu = _ops()
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd._lib import LLDWTError
    z = np.load(GOLDEN + "/cdf97_pywt_small.npz")
    x5 = torch.rand(1, 3, 2, 96, 160, generator=torch.Generator().manual_seed(5)) - 0.5     # 5 levels: down to 6 x 10 -> 3 x 5
    # default policy: a drop-in must not differ silently from the reference's single-fold form -> the call raises
    with pytest.raises(LLDWTError, match="shorter than the 10-tap"):
        ops.cdf97_forward(gu.dev(x5), 5)
    ll4, yh4 = ops.cdf97_forward(gu.dev(x5), 4)                                               # 12 x 20 at the last level: fine
    with pytest.raises(LLDWTError, match="shorter than the 10-tap"):
        ops.cdf97_inverse(ll4[..., :3, :5].contiguous(), yh4 + [ll4[..., None, :3, :5].expand(-1, -1, -1, 3, -1, -1).contiguous()])
    ops.set_cdf97_short_levels(True)                         # opt into the exact periodic transform
    try:
        for name in "abcd":
            x = torch.tensor(z[name + "_x"], dtype=torch.float32)
            lev = int(z[name + "_levels"])
            ll, yh = ops.cdf97_forward(gu.pm(x), lev)
            assert maxdiff(ll[0].cpu(), torch.tensor(z[name + "_ll"])) < 5e-5, name
            for i in range(lev):
                assert maxdiff(yh[i][0].cpu(), torch.tensor(z["%s_yh%d" % (name, i)])) < 5e-5, (name, i)
            assert maxdiff(ops.cdf97_inverse(ll, yh)[0].cpu(), x) < 5e-5, name
        ll, yh = ops.cdf97_forward(gu.dev(x5), 5)
        oll, oyh = cdf97.dwt_forward(x5[0], 5, periodic=True)
        assert maxdiff(ll[0].cpu(), oll) < 5e-5
        for i in range(5):
            assert maxdiff(yh[i][0].cpu(), oyh[i]) < 5e-5, i
        assert maxdiff(ops.cdf97_inverse(ll, yh).cpu(), x5) < 5e-5
    finally:
        ops.set_cdf97_short_levels(False)


@pytest.mark.parametrize("hw,levels", [((40, 48), 3), ((80, 96), 4), ((20, 12), 1)])
def test_cdf97_at_the_short_level_boundary_equals_the_reference_form(hw, levels):
    """The last level's input is exactly 10 or 12 samples on one side: the smallest sizes the default policy accepts.  There
    the kernels must equal the oracle's DEFAULT form (the single fold restated from pytorch_wavelets' afb1d, which is what
    the reference computes) -- and the periodic form, since the two coincide from 10 samples up."""
    ops, gu = _ops()
    x = torch.rand(1, 2, 1, *hw, generator=torch.Generator().manual_seed(hw[0])) - 0.5
    ll, yh = ops.cdf97_forward(gu.dev(x), levels)
    for periodic in (False, True):
        oll, oyh = cdf97.dwt_forward(x[0], levels, periodic=periodic)
        assert maxdiff(ll[0].cpu(), oll) < 5e-5, periodic
        for i in range(levels):
            assert maxdiff(yh[i][0].cpu(), oyh[i]) < 5e-5, (i, periodic)
    assert maxdiff(ops.cdf97_inverse(ll, yh).cpu(), x) < 5e-5
    assert maxdiff(cdf97.dwt_inverse(*cdf97.dwt_forward(x[0], levels)), x[0]) < 5e-5


@pytest.mark.parametrize("shape,levels", [((1, 2, 1, 68, 136), 1), ((1, 1, 2, 140, 72), 2), ((1, 1, 1, 64, 64), 1)])
def test_cdf97_fast_kernels_near_their_size_limit(shape, levels):
    """The fast tile kernels take levels from 64 samples (forward) / 32-sample subbands (inverse): the 72-wide patch of a
    tile is then wider than what is left of the image and wraps -- including a second, nearly empty tile (68 rows -> 34
    subband rows).  Compared with the oracle's periodic transform, and inverted."""
    ops, gu = _ops()
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[-1])) - 0.5
    ll, yh = ops.cdf97_forward(gu.dev(x), levels)
    oll, oyh = cdf97.dwt_forward(x[0], levels, periodic=True)
    assert maxdiff(ll[0].cpu(), oll) < 5e-5
    for i in range(levels):
        assert maxdiff(yh[i][0].cpu(), oyh[i]) < 5e-5, i
    assert maxdiff(ops.cdf97_inverse(ll, yh).cpu(), x) < 5e-5


def test_cdf97_tile_shapes_agree_across_batch_sizes():
    """The tile kernels choose their tile by a level's tile count: 32 x 32 from 4 096 square tiles per level up (large batches), 16 x 32
    below, 8 x 32 (forward) under 200 -- one arithmetic per sample, so a plane transformed inside a large batch (level 0 on square tiles,
    the deeper levels on 16 x 32) must equal the same plane transformed alone (16 x 32 / 8 x 32 tiles) BIT FOR BIT, both directions."""
    ops, gu = _ops()
    g = torch.Generator().manual_seed(97)
    big = gu.dev(torch.rand(1, 64, 1, 512, 512, generator=g) - 0.5)            # level 0: 64 * 8 * 8 = 4 096 square tiles
    one = big[:, 17:18].contiguous()
    ll_b, yh_b = ops.cdf97_forward(big, 4)
    ll_1, yh_1 = ops.cdf97_forward(one, 4)
    assert torch.equal(ll_b[:, 17:18], ll_1)
    for a, b in zip(yh_b, yh_1):
        assert torch.equal(a[:, 17:18], b)
    xb = ops.cdf97_inverse(ll_b, yh_b)
    x1 = ops.cdf97_inverse(ll_1, yh_1)
    assert torch.equal(xb[:, 17:18], x1)
    assert maxdiff(xb.cpu(), big.cpu()) < 5e-5
