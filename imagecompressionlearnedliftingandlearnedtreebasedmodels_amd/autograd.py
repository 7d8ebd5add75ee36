"""torch.autograd.Function wrappers: autograd is used as the tape (plumbing); every forward AND backward computation is a
HIP kernel behind the C-ABI (include/lldwt.h).  The reference gets its backward from PyTorch autograd over ATen ops
(agents/liftingDWT_agent.py:97); here the gradients are hand-written kernels:
  * backward-data of a conv  = the forward MFMA engine on the gradient with flipped taps / swapped channels
  * backward-weights         = lldwt_conv2d_wgrad (MFMA GEMM over pixels, split over images, float atomics)
  * activations              = fused into the producing kernel's epilogue or lldwt_act_bwd
  * Gaussian rate            = closed-form d/dx, d/dsigma, d/dmu with both LowerBound gradient rules
"""
import ctypes
import os

import torch

from . import ops


class ConvFn(torch.autograd.Function):
    """y = act(conv(x, w) + b [+ residual]); x (P,B,cin,h,w) [(P,B,cin,h/2,w/2) if upsample2], w (P,cout,cin/g,K,K)."""

    @staticmethod
    def _f16x3(x, w, K, groups, upsample2, tap_mask, residual):
        """Dense 3x3 convs with many channels (the 243 -> 243 tree conv) run on the fp16 matrix cores with split-fp16
        operands in training too -- forward and backward-data (same kernel, transposed + flipped weights); fp32-level
        accuracy (csrc/conv_f16x3.hip); the weight gradient likewise (csrc/conv_wgrad_f16x3.hip, rows that are multiples of 4)."""
        return (K == 3 and groups == 1 and not upsample2 and tap_mask is None and residual is None and w.shape[1] >= 64 and
                w.shape[2] >= 64 and ops.plc_mode() == "f16x3")

    @staticmethod
    def forward(ctx, x, w, b, residual, K, groups, act, upsample2, tap_mask):
        fast = ConvFn._f16x3(x, w, K, groups, upsample2, tap_mask, residual)
        xs = None
        if fast:
            xs = ops.absmax_slots(x)            # kept for the weight gradient: one |max| pass over x per step instead of two
            y = ops.conv3x3_f16x3(x, ops.conv_f16x3_pack(w.detach().contiguous()), b, w.shape[1], act=act, slots=xs)
        else:
            y = ops.conv2d(x, w, b, K, groups=groups, act=act, upsample2=upsample2, tap_mask=tap_mask, residual=residual)
        ctx.save_for_backward(x, w, y if act != ops.ACT_NONE else None)
        ctx.cfg = (K, groups, act, upsample2, tap_mask, b is not None, residual is not None)
        ctx.fast = fast
        ctx.x_slots = xs
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        K, groups, act, upsample2, tap_mask, has_b, has_res = ctx.cfg
        dy = dy.contiguous()
        dpre = ops.act_bwd(dy, y, act) if act != ops.ACT_NONE else dy
        dx = dw = db = None
        ds = ops.absmax_slots(dpre) if ctx.fast else None                      # shared by backward-data and backward-weights
        if ctx.needs_input_grad[0] and ctx.fast:
            wt = w.detach().transpose(1, 2).flip(-1, -2).contiguous()          # (P, cin, cout, 3, 3): the adjoint conv's weight
            dx = ops.conv3x3_f16x3(dpre, ops.conv_f16x3_pack(wt), None, w.shape[2], slots=ds)
        elif ctx.needs_input_grad[0]:
            # backward-data: the forward weight read as a ConvTranspose2d weight (cin' = cout) with flipped taps
            dx = ops.conv2d(dpre, w, None, K, groups=groups, transposed=True, tap_mask=ops.flip_mask(tap_mask, K))
            if upsample2:
                dx = ops.downsum2(dx)
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            if ctx.fast and x.shape[-1] % 4 == 0:
                dw, db = ops.conv3x3_wgrad_f16x3(x, dpre, tuple(w.shape), want_bias=has_b,      # split-fp16 matrix cores
                                                 x_slots=ctx.x_slots, dy_slots=ds)
            else:
                dw, db = ops.conv2d_wgrad(x, dpre, tuple(w.shape), K, groups=groups, upsample2=upsample2, tap_mask=tap_mask,
                                          want_bias=has_b)
        return dx, dw, db, (dpre if has_res else None), None, None, None, None, None


def conv(x, w, b, K, groups=1, act=ops.ACT_NONE, upsample2=False, tap_mask=None, residual=None):
    return ConvFn.apply(x, w, b, residual, K, groups, act, upsample2, tap_mask)


# ------------------------------------------------------------------------------------------------ rate / quantise / colour
class GaussRateFn(torch.autograd.Function):
    """bits = -log2 p(x + noise | sigma, mu) (lldwt_gauss_rate); backward closed form (lldwt_gauss_rate_bwd)."""

    @staticmethod
    def forward(ctx, x, params, noise):
        bits, _ = ops.gauss_rate(x, params, noise)
        ctx.save_for_backward(x, params, noise)
        return bits

    @staticmethod
    def backward(ctx, gbits):
        x, params, noise = ctx.saved_tensors
        dx, dparams = ops.gauss_rate_bwd(x, params, noise, gbits.contiguous())
        return dx, dparams, None


_MLP_WGRAD_FUSED = os.environ.get("LLDWT_MLP_WGRAD", "fused") != "gemm"
_CGP_TRAIN_F16 = os.environ.get("LLDWT_CGP_TRAIN_FWD", "f16x3") != "f32"
_CGP_TRAIN_BWD_F16 = os.environ.get("LLDWT_CGP_TRAIN_BWD", "f16x3") != "f32"


class SubbandMlpFn(torch.autograd.Function):
    """SubbandAutoEncoder MLP (1 -> 32 -> 32 -> 32 -> 1 per coefficient, grouped 1x1 convs) as one forward kernel and one
    backward-data kernel, both on the matrix cores with the activations in registers; the four weight gradients are the
    grouped 1x1 GEMMs.  Weights in Conv2d layout: w0 (P,C*32,1,1,1), w1,w2 (P,C*32,32,1,1), w3 (P,C,32,1,1)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, w2, b2, w3, b3):
        ctx.save_for_backward(x, w0, b0, w1, b1, w2, b2, w3)
        return ops.subband_mlp(x, w0, b0, w1, b1, w2, b2, w3, b3, transposed=False)

    @staticmethod
    def backward(ctx, gy):
        x, w0, b0, w1, b1, w2, b2, w3 = ctx.saved_tensors
        Cc = x.shape[2]
        gy = gy.contiguous()
        if _MLP_WGRAD_FUSED:
            # one launch: backward-data and the eight parameter gradients (nothing but x, gy, gx in HBM); LLDWT_MLP_WGRAD=gemm
            # keeps the form below (hidden activations and gradients written out, four grouped 1x1 weight-gradient GEMMs)
            gx, grads = ops.subband_mlp_bwd_w(x, gy, w0, b0, w1, b1, w2, b2, w3)
            return (gx, *grads)
        gx, hs, ds = ops.subband_mlp_bwd(x, gy, w0, b0, w1, b1, w2, b2, w3)
        grads = []
        for xin, dy, w in ((x, ds[0], w0), (hs[0], ds[1], w1), (hs[1], ds[2], w2), (hs[2], gy, w3)):
            dw, db = ops.conv2d_wgrad(xin, dy, tuple(w.shape), 1, groups=Cc)
            grads += [dw, db]
        return (gx, *grads)


class CgpRateFn(torch.autograd.Function):
    """bits of the fused cgp stack (four grouped 1x1 convs + Gaussian rate, LiftingBasedDWT_net.py:282-289,360-365) with a
    fused backward: lldwt_gauss_rate_bwd -> lldwt_cgp_bwd (all four backward-data passes in one launch) -> four 1x1
    weight-gradient GEMMs.  cat (P,B,G*c0,h,w); x, noise (P,B,G,h,w); w_l (P,G*c_{l+1},c_l,1,1); b_l (P,G*c_{l+1})."""

    @staticmethod
    def forward(ctx, cat, x, noise, groups, *wb):
        ws, bs = list(wb[0::2]), list(wb[1::2])
        packed, dims = ops.cgp_pack(ws, bs, groups)
        bits, params, h1, h2, h3 = ops.cgp_rate_train(cat, x, packed, dims, noise)
        ctx.save_for_backward(cat, x, noise, params, h1, h2, h3, *ws)
        ctx.dims, ctx.groups = dims, groups
        return bits

    @staticmethod
    def backward(ctx, gbits):
        cat, x, noise, params, h1, h2, h3, *ws = ctx.saved_tensors
        dims, G = ctx.dims, ctx.groups
        dx, dparams = ops.gauss_rate_bwd(x, params, noise, gbits.contiguous())
        dcat, d1, d2, d3 = ops.cgp_bwd(dparams, h1, h2, h3, ops.cgp_pack_bwd(ws, G), dims, G)
        grads = []
        for xin, dy, w in ((cat, d1, ws[0]), (h1, d2, ws[1]), (h2, d3, ws[2]), (h3, dparams, ws[3])):
            dw, db = ops.conv2d_wgrad(xin, dy, tuple(w.shape), 1, groups=G)
            grads += [dw, db]
        return (dcat, dx, None, None, *grads)


class CgpRateCtxFn(torch.autograd.Function):
    """CgpRateFn without the concatenated input: plc (P,B,G*cplc,h,w) = the tree-context conv's output, xq (P,B,G,h,w) = the
    quantised subband whose live causal taps (tap_mask of the K x K masked context conv, folded into layer 0 on the host) the
    kernels gather themselves.  Forward lldwt_cgp_rate_train_ctx; backward lldwt_cgp_bwd_split (input gradient as dplc + dtaps),
    layer 0's weight gradient from the two sources (lldwt_wgrad1x1_split), d(xq) = the taps' transpose (shifted adds).  The
    [plc_g | taps_g] tensor of CgpRateFn was 1.75 GB at the level-0 shape of configs[2]: a torch.cat in the forward, and the same
    copy again for the gradient of plc on the way back."""

    @staticmethod
    def _taps(K, tap_mask):
        return [t for t in range(K * K) if (tap_mask >> t) & 1]

    @staticmethod
    def _gather(xq, K, live):
        """patches[:, :, g*ntaps + j] = xq[:, :, g] shifted by live tap j, zero outside the image (as _fold_csc_train)."""
        import torch.nn.functional as F
        P, B, G, h, w = xq.shape
        R = K // 2
        xp = F.pad(xq, (R, R, R, R))
        taps = [xp[:, :, :, (t // K):(t // K) + h, (t % K):(t % K) + w] for t in live]
        return torch.stack(taps, dim=3).reshape(P, B, G * len(live), h, w).contiguous()

    @staticmethod
    def forward(ctx, plc, xq, x, noise, groups, K, tap_mask, *wb):
        ws, bs = list(wb[0::2]), list(wb[1::2])
        dims = tuple([ws[0].shape[2]] + [w_.shape[1] // groups for w_ in ws[:3]])
        if _CGP_TRAIN_F16 and dims == (93, 162, 54, 18) and ops.cgp_mode() == "f16x3":
            # the forward on the eval path's split-fp16 register chain, which also writes the hidden activations (fp32-level accuracy;
            # 1.4 -> ~3 ms per step against 6.0 for the fp32-MFMA kernel); LLDWT_CGP_TRAIN_FWD=f32 keeps the latter
            params, h1, h2, h3 = ops.cgp16_params_train(plc, xq, ops.cgp16_pack(ws, bs, groups), K, tap_mask)
            bits, _ = ops.gauss_rate(x, params, noise)
        else:
            packed, dims = ops.cgp_pack(ws, bs, groups)
            bits, params, h1, h2, h3 = ops.cgp_rate_train_ctx(plc, xq, x, packed, dims, noise, K, tap_mask)
        ctx.save_for_backward(plc, xq, x, noise, params, h1, h2, h3, *ws)
        ctx.dims, ctx.groups, ctx.K, ctx.tap_mask = dims, groups, K, tap_mask
        return bits

    @staticmethod
    def backward(ctx, gbits):
        plc, xq, x, noise, params, h1, h2, h3, *ws = ctx.saved_tensors
        dims, G, K = ctx.dims, ctx.groups, ctx.K
        live = CgpRateCtxFn._taps(K, ctx.tap_mask)
        nt, R = len(live), K // 2
        dx, dparams = ops.gauss_rate_bwd(x, params, noise, gbits.contiguous())
        if _CGP_TRAIN_BWD_F16 and dims == (93, 162, 54, 18) and nt == 12 and ops.cgp_mode() == "f16x3":
            # backward-data on the split-fp16 register chain (LLDWT_CGP_TRAIN_BWD=f32 keeps the fp32-MFMA kernel)
            dplc, dtaps, d1, d2, d3 = ops.cgp16_bwd(dparams, h1, h2, h3, ops.cgp16_pack_bwd(ws, G), G)
        else:
            dplc, dtaps, d1, d2, d3 = ops.cgp_bwd_split(dparams, h1, h2, h3, ops.cgp_pack_bwd(ws, G), dims, G, nt)
        grads = list(ops.wgrad1x1_split(plc, CgpRateCtxFn._gather(xq, K, live), d1, G))
        for xin, dy, w in ((h1, d2, ws[1]), (h2, d3, ws[2]), (h3, dparams, ws[3])):
            dw, db = ops.conv2d_wgrad(xin, dy, tuple(w.shape), 1, groups=G)
            grads += [dw, db]
        # d(xq): tap j of pixel p read xq at p + (dy_j - R, dx_j - R) -> its gradient is added there (transpose of the gather)
        P, B, _, h, w = xq.shape
        dq = torch.zeros(P, B, G, h + 2 * R, w + 2 * R, device=xq.device, dtype=torch.float32)
        dt = dtaps.view(P, B, G, nt, h, w)
        for j, t in enumerate(live):
            dq[:, :, :, (t // K):(t // K) + h, (t % K):(t % K) + w] += dt[:, :, :, j]
        dxq = dq[:, :, :, R:R + h, R:R + w].contiguous()
        return (dplc, dxq, dx, None, None, None, None, *grads)


class FactorizedRateFn(torch.autograd.Function):
    """(bits, q) of the factorized model (lldwt_factorized_rate); eb: (P,C,59) packed raw parameters (built by torch.cat
    from the module parameters, so their gradients flow back through the tape)."""

    @staticmethod
    def forward(ctx, x, eb, noise):
        bits, q = ops.factorized_rate(x, eb, noise)
        ctx.save_for_backward(x, eb, noise)
        ctx.mark_non_differentiable(q) if noise is None else None
        return bits, q

    @staticmethod
    def backward(ctx, gbits, gq):
        x, eb, noise = ctx.saved_tensors
        dx, deb = ops.factorized_rate_bwd(x, eb, noise, gbits.contiguous())
        if noise is not None and gq is not None:
            dx = dx + gq            # q = x + noise: identity path to the decoder
        return dx, deb, None


class Cdf97Fn(torch.autograd.Function):
    """Fixed CDF 9/7 analysis; backward = its adjoint (lldwt_cdf97_inverse_ex adj=1; bior4.4 is not orthogonal)."""

    @staticmethod
    def forward(ctx, x, levels):
        ll, yh = ops.cdf97_forward(x, levels)
        return (ll, *yh)

    @staticmethod
    def backward(ctx, g_ll, *g_yh):
        return ops.cdf97_inverse(g_ll.contiguous(), [t.contiguous() for t in g_yh], adj=True), None


class Cdf97InvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ll, *yh):
        ctx.levels = len(yh)
        return ops.cdf97_inverse(ll.contiguous(), [t.contiguous() for t in yh])

    @staticmethod
    def backward(ctx, gx):
        g_ll, g_yh = ops.cdf97_forward(gx.contiguous(), ctx.levels, adj=True)
        return (g_ll, *g_yh)


class SquareFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.ew_mul(x, x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.ew_mul(x, g.contiguous(), 2.0)


class GdnApplyFn(torch.autograd.Function):
    """y = x * rsqrt(nrm)  (inverse: x * sqrt(nrm)) -- graphs/layers/gdn.py:85-90."""

    @staticmethod
    def forward(ctx, x, nrm, inverse):
        ctx.save_for_backward(x, nrm)
        ctx.inverse = inverse
        return ops.gdn_apply(x, nrm, inverse)

    @staticmethod
    def backward(ctx, g):
        x, nrm = ctx.saved_tensors
        dx, dn = ops.gdn_apply_bwd(x, nrm, g.contiguous(), ctx.inverse)
        return dx, dn, None


class NonNegParamFn(torch.autograd.Function):
    """NonNegativeParametrizer: max(x, bound)^2 - pedestal with the LowerBound gradient rule (utils/parametrizers.py:45-48)."""

    @staticmethod
    def forward(ctx, x, minimum):
        ctx.save_for_backward(x)
        ctx.minimum = minimum
        return ops.nonneg_param_fwd(x.contiguous(), minimum)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.nonneg_param_bwd(x.contiguous(), g.contiguous(), ctx.minimum), None


def gdn_train(x, beta, gamma, inverse, beta_min):
    """Differentiable GDN on plane-major tensors: x (P,B,C,h,w), beta (P,C), gamma (P,C,C)."""
    P, C_ = beta.shape
    b = NonNegParamFn.apply(beta, beta_min)
    g = NonNegParamFn.apply(gamma, 0.0).reshape(P, C_, C_, 1, 1)
    nrm = conv(SquareFn.apply(x), g, b, 1)
    return GdnApplyFn.apply(x, nrm, inverse)


class QuantNoiseFn(torch.autograd.Function):
    """quantize(x, 'noise') = x + U(-.5,.5): identity gradient."""

    @staticmethod
    def forward(ctx, x, noise):
        return ops.quantize(x, noise)

    @staticmethod
    def backward(ctx, g):
        return g, None


class YccToRgbFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ycc):
        return ops.ycc_to_rgb(ycc, clamp=False)

    @staticmethod
    def backward(ctx, g):
        return ops.ycc_to_rgb_bwd(g.contiguous())


class SumFn(torch.autograd.Function):
    """float64 device sum of a tensor (lldwt_sum); gradient = broadcast."""

    @staticmethod
    def forward(ctx, t):
        acc = torch.zeros(1, dtype=torch.float64, device=t.device)
        ops.sum_into(t.contiguous(), acc)
        ctx.shape = t.shape
        return acc

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.float32).expand(ctx.shape).contiguous()


class SqErrSumFn(torch.autograd.Function):
    """sum((a-b)^2) in float64 (lldwt_sq_err_sum); d/db = 2 (b - a) g (a is the target, no gradient)."""

    @staticmethod
    def forward(ctx, a, b):
        acc = torch.zeros(1, dtype=torch.float64, device=a.device)
        ops.sq_err_sum(a.contiguous(), b.contiguous(), acc)
        ctx.save_for_backward(a, b)
        return acc

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        # 2 g (b - a) with g left on the device: float(g) stalled the host here until the whole forward had drained
        return None, ops.axpby(b.contiguous(), a.contiguous(), 2.0, -2.0).mul_(g.to(torch.float32))


# ------------------------------------------------------------------------------------------------ lifting transform
_BUF_X, _BUF_LROW, _BUF_HROW, _BUF_TMPL, _BUF_TMPH, _BUF_LL0, _BUF_LL1, _BUF_LL, _BUF_YH0 = range(9)


_W_KEYS = ("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4")
_PACK_MEMO = {"key": None, "val": None}


def _pack_forward(W, nblocks):
    """W: dict of 8 stacked tensors (nblocks,2,P,...) -> packed (P,nblocks,2,total) for the lifting step kernels.

    One training step asks for the same pack four times (encode and decode share their P/U blocks -- lifting_dwt_nets.py:
    695-707 -- and each backward reads the forward pack again): the last result is kept and reused while every tensor is
    the same object at the same version (an optimizer step or load_state_dict bumps ``_version``).  With the fused training
    forward the pack includes the composed 9x9 kernels (0.4 ms per block and call: 6.7 ms per step without the memo)."""
    key = tuple((k, W[k].data_ptr(), W[k]._version, tuple(W[k].shape)) for k in _W_KEYS) + (nblocks, ops.train_lift_f16())
    if _PACK_MEMO["key"] == key:
        return _PACK_MEMO["val"]
    blocks = []
    for b in range(nblocks):
        # re-packed every step.  The fused f16x3 training forward (default for 16 channels, 5x5, tanh) reads the split-fp16
        # section; with LLDWT_TRAIN_LIFT=f32 / LLDWT_LIFT_MODE=f32 only the fp32 kernels run and that section is skipped
        pu = [ops.pack_pblock(*[W[k][b, u] for k in ("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4")],
                              train=not ops.train_lift_f16(), compose=False) for u in range(2)]
        blocks.append(torch.stack(pu, 1))
    out = torch.stack(blocks, 1).contiguous()
    _PACK_MEMO["key"], _PACK_MEMO["val"] = key, out
    return out


_BPACK_MEMO = {"key": None, "val": None}


def _pack_backward(W, nblocks):
    """Backward packs (transposed, mirrored weights; ops.pack_pblock_bwd) of every P/U block, (P,nblocks,2,total) like
    _pack_forward's result, kept while the weights are unchanged (the encode and the decode backward of a step share it)."""
    key = tuple((k, W[k].data_ptr(), W[k]._version, tuple(W[k].shape)) for k in ("w1", "w2", "w3", "w4")) + (nblocks,)
    if _BPACK_MEMO["key"] == key:
        return _BPACK_MEMO["val"]
    out = torch.stack([torch.stack([ops.pack_pblock_bwd(*[W[k][b, u] for k in ("w1", "w2", "w3", "w4")]) for u in range(2)], 1)
                       for b in range(nblocks)], 1).contiguous()
    _BPACK_MEMO["key"], _BPACK_MEMO["val"] = key, out
    return out


class _LiftBackward:
    """Executes the step program in reverse over gradient buffers of the forward layout (include/lldwt.h)."""

    def __init__(self, meta, taps, W, saved, P, B, H, W_, nh=None, nl=None):
        self.m, self.taps, self.W, self.saved = meta, taps, W, saved
        self.nh, self.nl = nh, nl                                   # (P,) effective gains of config.scale == 1, or None
        self.dnh = torch.zeros_like(nh) if nh is not None else None
        self.dnl = torch.zeros_like(nl) if nl is not None else None
        self.P, self.B, self.H, self.Wd = P, B, H, W_
        self.Z = P * B
        self.dtaps = torch.zeros_like(taps)
        self.dW = {k: torch.zeros_like(v) for k, v in W.items()}
        self.packs = {}
        self.fwd_pack = _pack_forward(W, W["w1"].shape[0]) if meta["C"] == 16 else None     # (P,nblocks,2,total)
        # tanh block, 16 channels, 5x5: the backward-data chain runs on the fused split-fp16 kernel (LLDWT_BWD_LIFT=f32: fp32 MFMA)
        self.bwd_pack, self.taps_id = None, None
        if meta["C"] == 16 and meta["K"] == 5 and not meta["linear"] and ops.bwd_lift_f16():
            self.bwd_pack = _pack_backward(W, W["w1"].shape[0])
            self.taps_id = torch.tensor([0.0, 1.0, 0.0], device=taps.device).repeat(P, 1).contiguous()

    def _pack(self, name, blk, u, vertical):
        key = (name, blk, u, vertical)
        if key not in self.packs:
            self.packs[key] = ops.conv_pack(self.W[name][blk, u], self.m["K"], transposed=True, swap_hw=not vertical)
        return self.packs[key]

    def _scale_bwd(self, op, G, n):
        """Gain op of config.scale == 1 (wavelet_forward_v2.py:76-80, wavelet_inverse_v2.py:70-74): kind 1/2 y = v * s,
        kind 3/4 y = v / s with s = nh (1, 3) or nl (2, 4) per plane; v is what the forward kept in `saved`.
        G[src] = dL/dv (SET: an op's src view is either its own dout or read by nobody else), d(s) accumulates."""
        P, B, h, w = self.P, self.B, op.h, op.w
        s, ds = (self.nh, self.dnh) if op.kind in (1, 3) else (self.nl, self.dnl)
        if s is None:
            raise RuntimeError("lifting program with gain ops but no gains were given")
        v = self.saved[op.saved_off:op.saved_off + n].view(P, B, h, w)

        def gview(buf, off, sz, sy, sx):
            t = G[buf]
            return torch.as_strided(t, (P, B, h, w), (B * sz, sz, sy, sx), t.storage_offset() + off)
        gy = gview(op.buf_dout, op.off_dout, op.sz_dout, op.sy_dout, op.sx_dout)
        sv = s.view(P, 1, 1, 1)
        dot = (gy * v).sum(dim=(1, 2, 3))
        if op.kind <= 2:
            ds += dot
            gv = gy * sv
        else:
            ds -= dot / (s * s)
            gv = gy / sv
        gview(op.buf_src, op.off_src, op.sz_src, op.sy_src, op.sx_src).copy_(gv)

    def run(self, program, G):
        m, Z, B = self.m, self.Z, self.B
        C_, K, rw = m["C"], m["K"], m["rw"]
        epi = ops.EPI_NONE if m["linear"] else ops.EPI_TANH_BWD
        for op in reversed(program):
            h, w = op.h, op.w
            n = Z * h * w
            if op.kind != 0:
                self._scale_bwd(op, G, n)
                continue
            base = self.saved[op.saved_off:op.saved_off + n * (2 + 3 * C_)]
            srcv = base[:n]
            skip = base[n:2 * n].view(self.P, B, 1, h, w)
            t1 = base[2 * n:(2 + C_) * n].view(self.P, B, C_, h, w)
            t2 = base[(2 + C_) * n:(2 + 2 * C_) * n].view(self.P, B, C_, h, w)
            t3 = base[(2 + 2 * C_) * n:(2 + 3 * C_) * n].view(self.P, B, C_, h, w)
            view = lambda buf, off, sz, sy, sx: ops.View(G[buf].data_ptr() + 4 * off, sz, sy, sx)
            if self.fwd_pack is not None:
                # C == 16: the whole step backward is one C-ABI call (fused MFMA backward-data + dedicated wgrad kernels)
                blk, u = op.block, op.is_u
                nb_, tot = self.fwd_pack.shape[1], self.fwd_pack.shape[3]
                pk = ctypes.c_void_p(self.fwd_pack.data_ptr() + 4 * (blk * 2 + u) * tot)
                ops.lift_step_bwd(view(op.buf_dout, op.off_dout, op.sz_dout, op.sy_dout, op.sx_dout),
                                  view(op.buf_din, op.off_din, op.sz_din, op.sy_din, op.sx_din),
                                  view(op.buf_src, op.off_src, op.sz_src, op.sy_src, op.sx_src),
                                  base, self.P, B, h, w, self.taps[op.tap], self.dtaps[op.tap], pk, nb_ * 2 * tot,
                                  [self.dW[k][blk, u] for k in _W_KEYS], C_, K, rw, op.sign, bool(op.vertical),
                                  m["linear"],
                                  packed_bwd=None if self.bwd_pack is None else
                                  ctypes.c_void_p(self.bwd_pack.data_ptr() + 4 * (blk * 2 + u) * tot),
                                  taps_id=self.taps_id)
                continue
            g = torch.empty(self.P, B, 1, h, w, device=srcv.device, dtype=torch.float32)
            ops.lift_bwd_pre(view(op.buf_dout, op.off_dout, op.sz_dout, op.sy_dout, op.sx_dout),
                             view(op.buf_din, op.off_din, op.sz_din, op.sy_din, op.sx_din), g, Z, h, w)
            blk, u, vert = op.block, op.is_u, bool(op.vertical)
            Wb = {k: v[blk, u] for k, v in self.W.items()}
            dWb = {k: v[blk, u] for k, v in self.dW.items()}
            alpha = op.sign * rw
            swap = not vert
            # conv4: net = conv4(t3)
            dt3 = ops.conv2d(g, Wb["w4"], None, K, transposed=True, packed=self._pack("w4", blk, u, vert))
            ops.conv2d_wgrad(t3, g, tuple(Wb["w4"].shape), K, dw=dWb["w4"], db=dWb["b4"], alpha=alpha, swap_hw=swap)
            # conv3 (+ residual r): t3 = conv3(t2) + r
            dpre2 = ops.conv2d(dt3, Wb["w3"], None, K, transposed=True, packed=self._pack("w3", blk, u, vert), aux=t2,
                               epi=epi)
            ops.conv2d_wgrad(t2, dt3, tuple(Wb["w3"].shape), K, dw=dWb["w3"], db=dWb["b3"], alpha=alpha, swap_hw=swap)
            # conv2: t2 = tanh(conv2(t1)); dr = dt1 * tanh'(r) + dt3
            dr = ops.conv2d(dpre2, Wb["w2"], None, K, transposed=True, packed=self._pack("w2", blk, u, vert), aux=t1,
                            epi=epi, residual=dt3)
            ops.conv2d_wgrad(t1, dpre2, tuple(Wb["w2"].shape), K, dw=dWb["w2"], db=dWb["b2"], alpha=alpha, swap_hw=swap)
            # conv1: r = conv1(skip)
            dsk = ops.conv2d(dr, Wb["w1"], None, K, transposed=True, packed=self._pack("w1", blk, u, vert))
            ops.conv2d_wgrad(skip, dr, tuple(Wb["w1"].shape), K, dw=dWb["w1"], db=dWb["b1"], alpha=alpha, swap_hw=swap)
            ops.lift_bwd_fin(g, dsk, srcv, view(op.buf_src, op.off_src, op.sz_src, op.sy_src, op.sx_src), Z, B, h, w,
                             self.taps[op.tap], self.dtaps[op.tap],
                             vert, op.sign, rw)


def _grad_buffers(P, B, H, W, levels, dev):
    Z = P * B
    half, quarter = Z * (H // 2) * W, Z * (H // 2) * (W // 2)
    G = {}
    for b in (_BUF_LROW, _BUF_HROW, _BUF_TMPL, _BUF_TMPH):
        G[b] = torch.empty(half, device=dev, dtype=torch.float32)
    for b in (_BUF_LL0, _BUF_LL1):
        G[b] = torch.empty(quarter, device=dev, dtype=torch.float32)
    return G


class LiftingFn(torch.autograd.Function):
    """x (P,B,1,H,W) -> (ll, yh_0..yh_{L-1}); parameters: taps (4,P,3), the effective gains nh, nl (P,) of
    config.scale == 1 (None otherwise) and the 8 stacked P/U-block tensors (nblocks,2,P,...).
    Forward = lldwt_lifting_forward_train_ex; backward = reversed step program."""

    @staticmethod
    def forward(ctx, x, taps, meta, nh, nl, *Wt):
        W = dict(zip(_W_KEYS, Wt))
        P, B, _, H, Wd = x.shape
        nblocks = Wt[0].shape[0]
        scale = nh is not None
        prog, nsaved = ops.lifting_program(P * B, H, Wd, meta["levels"], meta["different"], 0, False, meta["C"], scale)
        saved = torch.empty(nsaved, device=x.device, dtype=torch.float32)
        packed = _pack_forward(W, nblocks)
        nh_, nl_ = (nh.detach().contiguous(), nl.detach().contiguous()) if scale else (None, None)
        ll, yh = ops.lifting_forward_train(x, taps, packed, meta["levels"], meta["C"], meta["K"], meta["rw"], meta["linear"],
                                           meta["different"], 0, saved, nh_, nl_)
        ctx.save_for_backward(taps, saved, *Wt)
        ctx.gains = (nh_, nl_)
        ctx.meta, ctx.prog, ctx.shape = meta, prog, (P, B, H, Wd)
        return (ll, *yh)

    @staticmethod
    def backward(ctx, g_ll, *g_yh):
        taps, saved, *Wt = ctx.saved_tensors
        W = dict(zip(_W_KEYS, Wt))
        P, B, H, Wd = ctx.shape
        L = ctx.meta["levels"]
        G = _grad_buffers(P, B, H, Wd, L, taps.device)
        G[_BUF_X] = torch.empty(P, B, 1, H, Wd, device=taps.device, dtype=torch.float32)
        G[_BUF_LL] = g_ll.contiguous().clone()
        for i in range(L):
            G[_BUF_YH0 + i] = g_yh[i].contiguous().clone()
        bw = _LiftBackward(ctx.meta, taps, W, saved, P, B, H, Wd, *ctx.gains)
        bw.run(ctx.prog, G)
        return (G[_BUF_X], bw.dtaps, None, bw.dnh, bw.dnl, *[bw.dW[k] for k in _W_KEYS])


class LiftingInvFn(torch.autograd.Function):
    """(ll, yh_0..yh_{L-1}) -> x (inverse transform); same parameters as LiftingFn."""

    @staticmethod
    def forward(ctx, taps, meta, nlev, nh, nl, *rest):
        ll, yh, Wt = rest[0], list(rest[1:1 + nlev]), rest[1 + nlev:]
        W = dict(zip(_W_KEYS, Wt))
        P, B, _, hl, wl = ll.shape
        H, Wd = hl << nlev, wl << nlev
        nblocks = Wt[0].shape[0]
        off = 2 * nlev if meta["different"] else 0     # lifting_dwt_nets.py:718-722
        scale = nh is not None
        prog, nsaved = ops.lifting_program(P * B, H, Wd, nlev, False, off, True, meta["C"], scale)
        saved = torch.empty(nsaved, device=ll.device, dtype=torch.float32)
        packed = _pack_forward(W, nblocks)
        nh_, nl_ = (nh.detach().contiguous(), nl.detach().contiguous()) if scale else (None, None)
        x = ops.lifting_inverse_train(ll.contiguous(), [t.contiguous() for t in yh], taps, packed, meta["C"], meta["K"],
                                      meta["rw"], meta["linear"], off, saved, nh_, nl_)
        ctx.save_for_backward(taps, saved, *Wt)
        ctx.gains = (nh_, nl_)
        ctx.meta, ctx.prog, ctx.shape, ctx.nlev = meta, prog, (P, B, H, Wd), nlev
        return x

    @staticmethod
    def backward(ctx, gx):
        taps, saved, *Wt = ctx.saved_tensors
        W = dict(zip(_W_KEYS, Wt))
        P, B, H, Wd = ctx.shape
        L = ctx.nlev
        G = _grad_buffers(P, B, H, Wd, L, taps.device)
        G[_BUF_X] = gx.contiguous().clone()
        G[_BUF_LL] = torch.empty(P, B, 1, H >> L, Wd >> L, device=taps.device, dtype=torch.float32)
        for i in range(L):
            G[_BUF_YH0 + i] = torch.empty(P, B, 3, H >> (i + 1), Wd >> (i + 1), device=taps.device, dtype=torch.float32)
        bw = _LiftBackward(ctx.meta, taps, W, saved, P, B, H, Wd, *ctx.gains)
        bw.run(ctx.prog, G)
        return (bw.dtaps, None, None, bw.dnh, bw.dnl, G[_BUF_LL], *[G[_BUF_YH0 + i] for i in range(L)],
                *[bw.dW[k] for k in _W_KEYS])
