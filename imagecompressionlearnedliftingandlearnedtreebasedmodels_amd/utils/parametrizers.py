"""NonNegativeParametrizer on the HIP path -- mirrors utils/parametrizers.py:23-48 of the reference."""
import torch
import torch.nn as nn

from .. import ops
from .bound_ops import LowerBound


class _NonNegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, minimum):
        ctx.save_for_backward(x)
        ctx.minimum = minimum
        return ops.nonneg_param_fwd(x.contiguous(), minimum)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.nonneg_param_bwd(x.contiguous(), g.contiguous(), ctx.minimum), None


class NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum=0, reparam_offset=2 ** -18):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = LowerBound((self.minimum + self.reparam_offset ** 2) ** 0.5)

    def init(self, x):
        # parameter initialisation only (host-side, tiny): sqrt(max(x + pedestal, pedestal)), parametrizers.py:42-43
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        # lower_bound(x)**2 - pedestal fused in one kernel (parametrizers.py:45-48)
        return _NonNegFn.apply(x, self.minimum)
