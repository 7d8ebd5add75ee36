"""torch.autograd.Function wrappers: autograd is used as the tape (plumbing); every forward AND backward computation is a
HIP kernel behind the C-ABI (include/lldwt.h).  The reference gets its backward from PyTorch autograd over ATen ops
(agents/liftingDWT_agent.py:97); here the gradients are hand-written kernels:
  * backward-data of a conv  = the forward MFMA engine on the gradient with flipped taps / swapped channels
  * backward-weights         = lldwt_conv2d_wgrad (MFMA GEMM over pixels, split over images, float atomics)
  * activations              = fused into the producing kernel's epilogue or lldwt_act_bwd
  * Gaussian rate            = closed-form d/dx, d/dsigma, d/dmu with both LowerBound gradient rules
"""
import torch

from . import ops


class ConvFn(torch.autograd.Function):
    """y = act(conv(x, w) + b [+ residual]); x (P,B,cin,h,w) [(P,B,cin,h/2,w/2) if upsample2], w (P,cout,cin/g,K,K)."""

    @staticmethod
    def forward(ctx, x, w, b, residual, K, groups, act, upsample2, tap_mask):
        y = ops.conv2d(x, w, b, K, groups=groups, act=act, upsample2=upsample2, tap_mask=tap_mask, residual=residual)
        ctx.save_for_backward(x, w, y if act != ops.ACT_NONE else None)
        ctx.cfg = (K, groups, act, upsample2, tap_mask, b is not None, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        K, groups, act, upsample2, tap_mask, has_b, has_res = ctx.cfg
        dy = dy.contiguous()
        dpre = ops.act_bwd(dy, y, act) if act != ops.ACT_NONE else dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # backward-data: the forward weight read as a ConvTranspose2d weight (cin' = cout) with flipped taps
            dx = ops.conv2d(dpre, w, None, K, groups=groups, transposed=True, tap_mask=ops.flip_mask(tap_mask, K))
            if upsample2:
                dx = ops.downsum2(dx)
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            dw, db = ops.conv2d_wgrad(x, dpre, tuple(w.shape), K, groups=groups, upsample2=upsample2, tap_mask=tap_mask,
                                      want_bias=has_b)
        return dx, dw, db, (dpre if has_res else None), None, None, None, None, None


def conv(x, w, b, K, groups=1, act=ops.ACT_NONE, upsample2=False, tap_mask=None, residual=None):
    return ConvFn.apply(x, w, b, residual, K, groups, act, upsample2, tap_mask)
