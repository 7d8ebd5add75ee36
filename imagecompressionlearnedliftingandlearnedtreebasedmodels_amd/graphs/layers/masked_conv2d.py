"""MaskedConv2d type A/B (reference graphs/layers/masked_conv2d.py:5-21) on the HIP conv kernels (dead taps skipped)."""
import torch.nn as nn

from ... import ops
from ...packed_cache import PackedOwnerMixin, invalidate_packed


class MaskedConv2d(PackedOwnerMixin, nn.Conv2d):
    def __init__(self, mask_type, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._init_packed_owner()
        assert mask_type in ("A", "B")
        self.mask_type = mask_type
        self.register_buffer("mask", self.weight.data.clone())
        _, _, kH, kW = self.weight.size()
        b = 1 if mask_type == "B" else 0
        self.mask.fill_(1)
        if kW > 1 or mask_type == "A":
            self.mask[:, :, kH // 2, kW // 2 + b:] = 0
        if kH > 1:
            self.mask[:, :, kH // 2 + 1:] = 0

    def tap_bits(self):
        """Bit t = ky*K+kx set for live taps (the mask is identical for every (out,in) pair)."""
        bits = getattr(self, "_tap_bits", None)
        if bits is None:        # the mask is a constant buffer: read it back from the device once, not on every forward
            m = self.mask[0, 0].flatten().tolist()
            bits = self._tap_bits = sum(1 << t for t, v in enumerate(m) if v > 0)
        return bits

    def apply_mask_(self):
        # the reference mutates weight.data in place on every forward (masked_conv2d.py:20).  Masking is idempotent, so
        # it is re-applied only when the weight tensor changed since the last masking (keeps the packed-weight cache hot)
        if getattr(self, "_masked_version", None) != (self.weight.data_ptr(), self.weight._version):
            self.weight.data *= self.mask          # a .data write: no version bump -> drop this module's packs explicitly
            invalidate_packed(self)
            self._masked_version = (self.weight.data_ptr(), self.weight._version)

    def forward(self, x):
        self.apply_mask_()
        K = self.kernel_size[0]
        y = ops.conv2d(x[None].contiguous(), self.weight.detach()[None].contiguous(),
                       None if self.bias is None else self.bias.detach()[None].contiguous(), K, groups=self.groups,
                       tap_mask=self.tap_bits())
        return y[0]
