"""GDN / inverse GDN (reference graphs/layers/gdn.py:41-92, vendored from compressai) on lldwt_gdn."""
import torch
import torch.nn as nn

from ... import ops
from ...utils.parametrizers import NonNegativeParametrizer

__all__ = ["GDN"]


class GDN(nn.Module):
    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_min = float(beta_min)
        self.beta_reparam = NonNegativeParametrizer(minimum=self.beta_min)
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def forward(self, x):
        """y = x * rsqrt(beta' + gamma' . x^2) (sqrt if inverse); re-parametrisation applied in-kernel (gdn.py:77-92)."""
        return ops.gdn(x[None].contiguous(), self.beta.detach()[None].contiguous(),
                       self.gamma.detach()[None].contiguous(), self.inverse, self.beta_min)[0]
