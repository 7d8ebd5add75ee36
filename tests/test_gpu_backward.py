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
    (486, 162, 1, 3, 2, False, None),      # cgp 1x1 grouped
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


def test_conv_backward_residual_and_large():
    """Residual branch gradient + a spatial size that is not a multiple of the 8x16 wgrad chunk."""
    ag, ops, gu = _mods()
    g = torch.Generator().manual_seed(5)
    P, B, C, K, h, w = 1, 2, 16, 5, 21, 37
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
