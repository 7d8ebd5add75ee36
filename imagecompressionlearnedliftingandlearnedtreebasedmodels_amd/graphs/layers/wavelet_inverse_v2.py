"""wavelet_inverse_v2 -- one level of the learned lifting, synthesis side (reference graphs/layers/wavelet_inverse_v2.py).

The merge (``reconstruct_fun``, :40-56) is pure addressing in the HIP kernels: the last two lifting steps write straight
into the even / odd rows (columns) of the output.
"""
import torch
import torch.nn as nn


class wavelet_inverse_v2(nn.Module):
    def __init__(self, P, U, resnet_coeff, liftingLevel, convBlockList, cfg, nh=0, nl=0, owner=None, level=0):
        super().__init__()
        self.P = P
        self.U = U
        self.lifting_level = liftingLevel
        self.resnet_coeff = resnet_coeff
        self.convBlock = convBlockList
        self.csize = cfg.clrch
        self.scale = cfg.scale
        self.nh = nh
        self.nl = nl
        self.config = cfg
        self._owner = [owner]
        self._level = level

    def one_level_lifting(self, LL, LH, HL, HH):
        """4 x (B,1,h/2,w/2) -> (B,1,h,w) (wavelet_inverse_v2.py:20-38)."""
        from .lifting_dwt_nets import lifting_inverse_planes
        yh = torch.cat((LH, HL, HH), 1)[None].contiguous()
        return lifting_inverse_planes([self._owner[0]], LL[None].contiguous(), [yh], first_level=self._level)[0]
