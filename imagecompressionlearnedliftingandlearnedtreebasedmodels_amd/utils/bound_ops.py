"""LowerBound on the HIP path -- mirrors utils/bound_ops.py:22-65 of the reference (same class / function names).

forward: max(x, bound); backward: g * [(x >= bound) | (g < 0)]  (utils/bound_ops.py:26-28), both HIP kernels
(lldwt_lower_bound_fwd / _bwd in include/lldwt.h).
"""
import torch
import torch.nn as nn

from .. import ops


def lower_bound_fwd(x, bound):
    return ops.lower_bound_fwd(x.contiguous(), float(bound))


def lower_bound_bwd(x, bound, grad_output):
    return ops.lower_bound_bwd(x.contiguous(), grad_output.contiguous(), float(bound)), None


class LowerBoundFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return lower_bound_fwd(x, bound)

    @staticmethod
    def backward(ctx, grad_output):
        x, bound = ctx.saved_tensors
        return lower_bound_bwd(x, bound, grad_output)


class LowerBound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return LowerBoundFunction.apply(x, self.bound)
