"""wavelet_forward_v2 -- one level of the learned lifting, analysis side (reference graphs/layers/wavelet_forward_v2.py).

Holds references to the shared P/U blocks, skip filters and gains exactly like the reference (so the state_dict carries the
same ``waveletForward.{l}.*`` aliases); ``one_level_lifting`` runs on the HIP lifting kernels.
"""
import torch.nn as nn


class wavelet_forward_v2(nn.Module):
    def __init__(self, P, U, resnet_coeff, liftingLevel, convBlockList, cfg, nh=0, nl=0, owner=None, level=0):
        super().__init__()
        self.P = P
        self.U = U
        self.resnet_weight = resnet_coeff
        self.lifting_level = liftingLevel
        self.csize = cfg.clrch
        self.convBlock = convBlockList
        self.nh = nh
        self.nl = nl
        self.scale = cfg.scale
        self._owner = [owner]      # list: keep the parent out of the module tree
        self._level = level

    def one_level_lifting(self, x):
        """(B,1,h,w) -> (LL, LH, HL, HH), HL = vertical-low / horizontal-high (wavelet_forward_v2.py:26-54)."""
        from .lifting_dwt_nets import lifting_forward_planes
        ll, yh = lifting_forward_planes([self._owner[0]], x[None].contiguous(), levels=1, first_level=self._level)
        return ll[0], yh[0][0][:, 0:1], yh[0][0][:, 1:2], yh[0][0][:, 2:3]
