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
    if act == 3:
        y = F.relu(y)
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
                                   (1, 1, 64, 64, 8, 32, 2), (1, 2, 64, 64, 12, 40, 3)])
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


def _pair64(parent, w1, b1, w2, b2, act):
    up = parent.double().repeat_interleave(2, dim=-2).repeat_interleave(2, dim=-1)
    t = F.leaky_relu(F.conv2d(up, w1.double(), b1.double(), padding=1), 0.01)
    return _ref64(t, w2, b2, act)


def test_fused_pair_operand_maps_exact_on_integers():
    """The one-launch tree-context pair (lldwt_plc_fused) on small integers, where every product and sum is exact: a wrong
    im2col index, channel row, halo pixel or padding rule shows as a wrong integer.  ReLU-free check: LeakyReLU multiplies
    negatives by 0.01, so the first conv is made non-negative (weights, parent, bias >= 0)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    P, B, cmid, cout, hp, wp = 2, 2, 45, 70, 7, 19                     # output 14 x 38: ragged tiles both ways
    parent = torch.randint(0, 4, (P, B, 3, hp, wp), generator=g).float()
    w1 = torch.randint(0, 3, (P, cmid, 3, 3, 3), generator=g).float()
    b1 = torch.randint(0, 4, (P, cmid), generator=g).float()
    w2 = torch.randint(-2, 3, (P, cout, cmid, 3, 3), generator=g).float()
    b2 = torch.randint(-5, 6, (P, cout), generator=g).float()
    pk1 = ops.plc_fused_pack1(w1.to(DEV), b1.to(DEV))
    pk2 = ops.conv_f16x3_pack(w2.to(DEV))
    y = ops.plc_fused(parent.to(DEV), pk1, pk2, b2.to(DEV), cmid, cout)
    for p in range(P):
        ref = _pair64(parent[p], w1[p], b1[p], w2[p], b2[p], 0)
        assert torch.equal(y[p].cpu().double(), ref)


@pytest.mark.parametrize("shape", [(1, 2, 243, 243, 32, 48, 0), (3, 1, 243, 243, 20, 17, 2), (1, 1, 100, 130, 4, 40, 0),
                                   (2, 1, 243, 243, 1, 1, 2)])
def test_fused_pair_accuracy_is_fp32_level(shape):
    """lldwt_plc_fused vs float64 and vs the two-launch fp32 engine: same 2e-6 bar as the unfused split-fp16 kernel, at
    tiny and large parent scales too (per-workgroup power-of-two scales)."""
    ops = _ops()
    P, B, cmid, cout, hp, wp, act = shape
    g = torch.Generator().manual_seed(cmid + hp)
    parent = (torch.rand(P, B, 3, hp, wp, generator=g) - 0.4) * 6.0
    w1 = (torch.rand(P, cmid, 3, 3, 3, generator=g) - 0.5) * 0.6
    b1 = torch.rand(P, cmid, generator=g) - 0.5
    w2 = (torch.rand(P, cout, cmid, 3, 3, generator=g) - 0.5) * (2.0 / (cmid * 9) ** 0.5)
    b2 = torch.rand(P, cout, generator=g) - 0.5
    pk1 = ops.plc_fused_pack1(w1.to(DEV), b1.to(DEV))
    pk2 = ops.conv_f16x3_pack(w2.to(DEV))
    for s in (1.0, 2.0 ** -18, 1.3e3):
        y = ops.plc_fused((parent * s).to(DEV), pk1, pk2, b2.to(DEV), cmid, cout, act=act)
        t32 = ops.conv2d((parent * s).to(DEV), w1.to(DEV), b1.to(DEV), 3, act=2, upsample2=True)
        y32 = ops.conv2d(t32, w2.to(DEV), b2.to(DEV), 3, act=act)
        assert y.shape == y32.shape
        for p in range(P):
            ref = _pair64(parent[p] * s, w1[p], b1[p], w2[p], b2[p], act)
            scale = float(ref.abs().max())
            e16 = float((y[p].cpu().double() - ref).abs().max()) / scale
            e32 = float((y32[p].cpu().double() - ref).abs().max()) / scale
            assert e16 < 2e-6, (s, e16, e32)
            assert e16 < 4 * e32 + 3e-7, (s, e16, e32)
    z = ops.plc_fused(torch.zeros_like(parent).to(DEV), pk1, pk2, b2.to(DEV), cmid, cout)       # all-zero parent
    t = F.leaky_relu(b1[0].double(), 0.01)[None, :, None, None].expand(1, cmid, 3, 3)
    centre = F.conv2d(t, w2[0].double(), b2[0].double())[0, :, 0, 0]
    if 2 * hp >= 3 and 2 * wp >= 3:
        assert float((z[0, 0, :, 1, 1].cpu().double() - centre).abs().max()) < 2e-6 * float(centre.abs().max())


def test_fused_pair_is_what_the_model_runs(monkeypatch):
    """_plc_pair: LLDWT_PLC_FUSE=1 (default) and =0 agree to fp32 level on an entropy layer's tree-context pair."""
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models import LiftingBasedDWT_net as net
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    ops = _ops()
    torch.manual_seed(2)
    layer = net.onlyEZWT(make_config(dwtlevels=2)).to(DEV).eval()
    seqs = [layer.plc_list[0]]
    parent = torch.randn(1, 2, 3, 24, 40, device=DEV)
    calls = []
    real = ops.plc_fused
    monkeypatch.setattr(ops, "plc_fused", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        monkeypatch.setenv("LLDWT_PLC_FUSE", "1")
        a = net._plc_pair([s[0] for s in seqs], [s[2] for s in seqs], parent, ops.ACT_NONE)
        monkeypatch.setenv("LLDWT_PLC_FUSE", "0")
        b = net._plc_pair([s[0] for s in seqs], [s[2] for s in seqs], parent, ops.ACT_NONE)
    assert len(calls) == 1
    assert float((a - b).abs().max()) < 2e-6 * float(b.abs().max())


@pytest.mark.parametrize("shape", [(1, 2, 243, 243, 32, 64), (2, 3, 70, 150, 17, 44), (1, 1, 64, 64, 2, 4), (1, 2, 96, 130, 9, 100)])
def test_wgrad_f16x3_vs_float64_and_fp32_kernel(shape):
    """lldwt_conv3x3_wgrad_f16x3 (split-fp16 MFMA GEMM over pixels, K split over workgroups, atomics) vs the float64
    gradient of F.conv2d and vs the fp32 MFMA weight-gradient kernel: ragged channel counts (partial oc / ic blocks), images
    smaller than a chunk, odd row counts, several images (the reduction runs over them), bias gradient."""
    ops = _ops()
    P, B, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(cin + h)
    x = (torch.rand(P, B, cin, h, w, generator=g) - 0.3) * 2.0
    dy = torch.randn(P, B, cout, h, w, generator=g) * 1e-3          # gradients are small numbers: the scales must cope
    dw, db = ops.conv3x3_wgrad_f16x3(x.to(DEV), dy.to(DEV), (P, cout, cin, 3, 3))
    dw32, db32 = ops.conv2d_wgrad(x.to(DEV), dy.to(DEV), (P, cout, cin, 3, 3), 3)
    for p in range(P):
        wz = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        bz = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x[p].double(), wz, bz, padding=1)
        (y * dy[p].double()).sum().backward()
        scale = float(wz.grad.abs().max())
        e16 = float((dw[p].cpu().double() - wz.grad).abs().max()) / scale
        e32 = float((dw32[p].cpu().double() - wz.grad).abs().max()) / scale
        assert e16 < 5e-6, (e16, e32)
        assert e16 < 4 * e32 + 5e-7, (e16, e32)
        eb = float((db[p].cpu().double() - bz.grad).abs().max()) / float(bz.grad.abs().max())
        assert eb < 5e-6, eb
    # accumulation semantics: alpha, and += into an existing buffer is the caller's (zeroed here)
    dw2, _ = ops.conv3x3_wgrad_f16x3(x.to(DEV), dy.to(DEV), (P, cout, cin, 3, 3), want_bias=False, alpha=-0.5)
    assert float((dw2 + 0.5 * dw).abs().max()) < 1e-5 * float(dw.abs().max())
    # the |max| slots handed in by the caller (lldwt_conv3x3_wgrad_f16x3_ex): same scales, same arithmetic -- equal up to the order of
    # the float atomics; either one alone, too
    xs, ds = ops.absmax_slots(x.to(DEV)), ops.absmax_slots(dy.to(DEV))
    for kw in (dict(x_slots=xs, dy_slots=ds), dict(x_slots=xs), dict(dy_slots=ds)):
        dw3, db3 = ops.conv3x3_wgrad_f16x3(x.to(DEV), dy.to(DEV), (P, cout, cin, 3, 3), **kw)
        assert float((dw3 - dw).abs().max()) < 1e-5 * float(dw.abs().max())
        assert float((db3 - db).abs().max()) < 1e-5 * float(db.abs().max())
    with pytest.raises(Exception):
        ops.conv3x3_wgrad_f16x3(x.to(DEV), dy.to(DEV), (P, cout, cin, 3, 3), x_slots=xs[:, :32].contiguous())


def test_wgrad_f16x3_both_kernels_agree():
    """k_wgrad3_f16x3_v2 (one copy of the input rows + register shifts, staging in the MFMA shadow; the default) against the first
    kernel (LLDWT_WGRAD3=v1, read at the first call: a child process) on a shape with several chunks per slice, partial channel
    blocks and a ragged last chunk column."""
    import subprocess, sys, os
    code = (
        "import torch, sys; sys.path.insert(0, %r)\n"
        "from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops\n"
        "g = torch.Generator().manual_seed(5)\n"
        "x = ((torch.rand(2, 3, 100, 38, 84, generator=g) - 0.3) * 2).cuda(); dy = (torch.randn(2, 3, 243, 38, 84, generator=g) * 1e-3).cuda()\n"
        "dw, db = ops.conv3x3_wgrad_f16x3(x, dy, (2, 243, 100, 3, 3))\n"
        "torch.save((dw.cpu(), db.cpu()), sys.argv[1])\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for mode in ("v1", "v2"):
            env = dict(os.environ, LLDWT_WGRAD3=mode)
            f = os.path.join(td, mode + ".pt")
            subprocess.run([sys.executable, "-c", code, f], check=True, env=env, timeout=300)
            out[mode] = torch.load(f, weights_only=True)
    for a, b in zip(out["v1"], out["v2"]):
        assert float((a - b).abs().max()) < 2e-6 * float(b.abs().max())


def test_fused_pair_random_shapes():
    """lldwt_plc_fused vs the two-launch fp32 engine on 16 seeded random shapes: parent sizes from 1x1 up (tiles overhanging
    the image both ways), channel counts that leave partial chunks / partial output blocks, several planes and images."""
    import random
    ops = _ops()
    rnd = random.Random(99)
    g = torch.Generator().manual_seed(8)
    worst = 0.0
    for case in range(16):
        P, B = rnd.randint(1, 2), rnd.randint(1, 3)
        cmid, cout = rnd.choice([17, 64, 100, 243, 256]), rnd.choice([5, 40, 129, 243])
        hp, wp = rnd.randint(1, 40), rnd.randint(1, 70)
        act = rnd.choice([0, 2, 1])
        parent = (torch.rand(P, B, 3, hp, wp, generator=g) - 0.5) * 4.0
        w1 = (torch.rand(P, cmid, 3, 3, 3, generator=g) - 0.5) * 0.6
        b1 = torch.rand(P, cmid, generator=g) - 0.5
        w2 = (torch.rand(P, cout, cmid, 3, 3, generator=g) - 0.5) * (2.0 / (cmid * 9) ** 0.5)
        b2 = torch.rand(P, cout, generator=g) - 0.5
        y = ops.plc_fused(parent.to(DEV), ops.plc_fused_pack1(w1.to(DEV), b1.to(DEV)), ops.conv_f16x3_pack(w2.to(DEV)),
                          b2.to(DEV), cmid, cout, act=act)
        t32 = ops.conv2d(parent.to(DEV), w1.to(DEV), b1.to(DEV), 3, act=2, upsample2=True)
        y32 = ops.conv2d(t32, w2.to(DEV), b2.to(DEV), 3, act=act)
        d = float((y - y32).abs().max()) / max(float(y32.abs().max()), 1e-6)
        worst = max(worst, d)
        assert d < 6e-6, (case, P, B, cmid, cout, hp, wp, act, d)      # two fp32-accurate paths: a few 1e-6 apart
    print("\n[plc_fused] vs the fp32 two-launch engine over 16 random shapes: worst relative difference %.3g" % worst)


def test_shape_flag_follows_the_environment():
    import os
    assert _ops().plc_shape() == (16 if os.environ.get("LLDWT_PLC_SHAPE") == "16" else 32)


def test_the_16x16x32_kernels_pass_this_file():
    """LLDWT_PLC_SHAPE=16 packs the weights for, and launches, the v_mfma_f32_16x16x32_f16 variants of the conv kernels (a process
    wide choice read when the library loads; the default 32x32x16 is the faster one on this path).  One child process runs this
    file's forward tests with it."""
    import os
    import subprocess
    import sys
    if os.environ.get("LLDWT_PLC_SHAPE"):
        pytest.skip("already inside the child run")
    env = dict(os.environ, LLDWT_PLC_SHAPE="16")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "operand_maps or accuracy_is_fp32 or random_shapes or shape_flag"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
