"""GPU: the split-fp16 dense 3x3 conv (csrc/conv_f16x3.hip) vs float64 convolution, and vs the fp32 MFMA engine."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    return ops


def _ref64(x, w, b, act):
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if act == 2:
        y = F.leaky_relu(y, 0.01)
    return y


def test_operand_maps_exact_on_integers():
    """Small integers are exact in fp16 and in the fp32 accumulator: any swapped row/column/k map shows as a wrong
    integer (asymmetric weights and inputs, cin and cout not multiples of the tiles, ragged image)."""
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    P, B, cin, cout, h, w = 2, 2, 37, 150, 11, 45
    x = torch.randint(-4, 5, (P, B, cin, h, w), generator=g).float()
    wt = torch.randint(-3, 4, (P, cout, cin, 3, 3), generator=g).float()
    b = torch.randint(-5, 6, (P, cout), generator=g).float()
    packed = ops.conv_f16x3_pack(wt.to(DEV))
    y = ops.conv3x3_f16x3(x.to(DEV), packed, b.to(DEV), cout)
    for p in range(P):
        ref = F.conv2d(x[p], wt[p], b[p], padding=1)
        assert torch.equal(y[p].cpu(), ref)


@pytest.mark.parametrize("shape", [(1, 2, 243, 243, 64, 96, 0), (2, 1, 243, 243, 40, 33, 2), (1, 1, 96, 192, 24, 70, 0),
                                   (1, 1, 64, 64, 8, 32, 2)])
def test_accuracy_is_fp32_level(shape):
    """Relative error vs a float64 reference is at the fp32 engine's level (~1e-6 of the output scale): the split keeps
    22 bits per operand.  Also: the result does not depend on the input's overall scale (power-of-two scaling)."""
    ops = _ops()
    P, B, cin, cout, h, w, act = shape
    g = torch.Generator().manual_seed(cin + h)
    x = (torch.rand(P, B, cin, h, w, generator=g) - 0.4) * 3.0
    wt = (torch.rand(P, cout, cin, 3, 3, generator=g) - 0.5) * (2.0 / (cin * 9) ** 0.5)
    b = torch.rand(P, cout, generator=g) - 0.5
    packed = ops.conv_f16x3_pack(wt.to(DEV))
    y = ops.conv3x3_f16x3(x.to(DEV), packed, b.to(DEV), cout, act=act)
    y32 = ops.conv2d(x.to(DEV), wt.to(DEV), b.to(DEV), 3, act=act)
    for p in range(P):
        ref = _ref64(x[p], wt[p], b[p], act)
        scale = float(ref.abs().max())
        e16 = float((y[p].cpu().double() - ref).abs().max()) / scale
        e32 = float((y32[p].cpu().double() - ref).abs().max()) / scale
        assert e16 < 2e-6, (e16, e32)
        assert e16 < 4 * e32 + 2e-7, (e16, e32)           # within a small factor of the exact-f32 fmaf chain
    for s in (2.0 ** -20, 3.7e4):                         # tiny and large activations: same relative accuracy
        ys = ops.conv3x3_f16x3((x * s).to(DEV), packed, torch.zeros_like(b).to(DEV), cout)
        for p in range(P):
            ref = F.conv2d((x[p] * s).double(), wt[p].double(), None, padding=1)
            assert float((ys[p].cpu().double() - ref).abs().max()) / float(ref.abs().max()) < 2e-6
    z = ops.conv3x3_f16x3(torch.zeros_like(x).to(DEV), packed, b.to(DEV), cout)      # all-zero input: scale 1, bias only
    assert torch.equal(z[0, 0, :, 0, 0].cpu(), b[0])


def test_absmax_slots():
    ops = _ops()
    x = torch.randn(3, 5, 7, 16, device=DEV)
    x[1, 2, 3, 4] = -77.5
    s = ops.absmax_slots(x)
    assert s.shape == (3, 64)
    assert torch.equal(s.max(dim=1).values.cpu(), x.abs().amax(dim=(1, 2, 3)).cpu())
