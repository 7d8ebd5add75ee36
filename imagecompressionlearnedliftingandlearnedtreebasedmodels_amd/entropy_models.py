"""Host-side mirrors of the two compressai==1.2.1 entropy models the reference uses (``EntropyBottleneck``,
``GaussianConditional``; imported at graphs/models/LiftingBasedDWT_net.py:3), running on the HIP rate kernels.

Parameter / buffer names and shapes follow compressai so reference checkpoints load.  The arithmetic of
rate estimation (likelihood + both LowerBounds + -log2) is one fused kernel: lldwt_gauss_rate / lldwt_factorized_rate
(include/lldwt.h).  The CDF tables of real entropy coding (``_offset``, ``_quantized_cdf``, ``_cdf_length``,
``scale_table``) are built by ``update_scale_table`` / ``update`` exactly as compressai 1.2.1 does (float32 pmf ->
``pmf_to_quantized_cdf`` at 16 bits) and consumed by ans.py / graphs/models/entropy_coding.py.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .utils.bound_ops import LowerBound

EB_ORDER = ("m0", "b0", "f0", "m1", "b1", "f1", "m2", "b2", "f2", "m3", "b3", "f3", "m4", "b4", "median")


def pack_entropy_bottleneck(sd):
    """{'_matrix0'.., '_bias0'.., '_factor0'.., 'quantiles'} of a C-channel EntropyBottleneck -> (C,59) raw values in
    the order the kernel expects (include/lldwt.h LLDWT_EB_FLOATS)."""
    C = sd["_matrix0"].shape[0]
    parts = []
    for i in range(5):
        parts.append(sd["_matrix%d" % i].reshape(C, -1))
        parts.append(sd["_bias%d" % i].reshape(C, -1))
        if i < 4:
            parts.append(sd["_factor%d" % i].reshape(C, -1))
    parts.append(sd["quantiles"][:, :, 1].reshape(C, 1))
    out = torch.cat(parts, 1).contiguous()
    assert out.shape[1] == 59
    return out


class _EntropyModelBase(nn.Module):
    def __init__(self, likelihood_bound=1e-9):
        super().__init__()
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self.likelihood_bound = float(likelihood_bound)
        self.likelihood_lower_bound = LowerBound(likelihood_bound)     # state_dict: likelihood_lower_bound.bound

    def _load_from_state_dict(self, state_dict, prefix, *args, **kw):
        # the CDF buffers change size with update(): take the incoming shapes (compressai: update_registered_buffers)
        for name in ("_offset", "_quantized_cdf", "_cdf_length", "scale_table"):
            key = prefix + name
            buf = getattr(self, name, None)
            if key in state_dict and buf is not None and tuple(state_dict[key].shape) != tuple(buf.shape):
                setattr(self, name, torch.empty(state_dict[key].shape, dtype=buf.dtype, device=buf.device))
        super()._load_from_state_dict(state_dict, prefix, *args, **kw)

    # compressai exposes the tables under these names (read by compress_ar, LiftingBasedDWT_net.py:463-465)
    @property
    def offset(self):
        return self._offset

    @property
    def quantized_cdf(self):
        return self._quantized_cdf

    @property
    def cdf_length(self):
        return self._cdf_length

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        """EntropyModel._pmf_to_cdf: per row, (pmf[:length], tail mass) -> quantised CDF at 16 bits."""
        from .ans import pmf_to_quantized_cdf
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[:int(pmf_length[i])], tail_mass[i]), dim=0)
            c = pmf_to_quantized_cdf(prob, 16)
            cdf[i, :len(c)] = torch.tensor(c, dtype=torch.int32)
        return cdf

    def quantize(self, inputs, mode, means=None):
        """compressai EntropyModel.quantize: 'noise' -> x + U(-.5,.5); 'dequantize' -> round(x - mu) + mu;
        'symbols' -> round(x - mu) as int32."""
        x = inputs.contiguous()
        if mode == "noise":
            noise = torch.empty_like(x).uniform_(-0.5, 0.5)
            return ops.quantize(x, noise)
        if mode not in ("dequantize", "symbols"):
            raise ValueError("unknown quantisation mode %r" % mode)
        r = ops.quantize(x if means is None else (x - means).contiguous())
        if mode == "symbols":
            return r.int()
        return r if means is None else r + means

    @staticmethod
    def dequantize(inputs, means=None):
        """symbols (int) -> float values, + means."""
        out = inputs.float()
        return out if means is None else out + means


class GaussianConditional(_EntropyModelBase):
    """compressai.entropy_models.GaussianConditional(scale_table=None, scale_bound=0.11) (LiftingBasedDWT_net.py:291)."""

    def __init__(self, scale_table=None, scale_bound=0.11, tail_mass=1e-9, **kw):
        super().__init__(**kw)
        self.tail_mass = float(tail_mass)
        if abs(float(scale_bound) - 0.11) > 1e-12:
            raise ValueError("the HIP rate kernel is built for scale_bound=0.11 (LiftingBasedDWT_net.py:291,307,318)")
        self.register_buffer("scale_table", torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]))
        self.lower_bound_scale = LowerBound(scale_bound)                # state_dict: lower_bound_scale.bound

    @staticmethod
    def _standardized_quantile(quantile):
        from scipy.stats import norm
        return float(norm.ppf(quantile))

    def update_scale_table(self, scale_table, force=False):
        """compressai GaussianConditional.update_scale_table (call site LiftingBasedDWT_net.py:462)."""
        if self._offset.numel() > 0 and not force and self.scale_table.numel() == len(scale_table) and \
                torch.equal(self.scale_table.cpu(), torch.as_tensor(scale_table).float().cpu()):
            return False
        dev = self.scale_bound.device
        self.scale_table = torch.as_tensor(scale_table).float().to(dev)
        self.update()
        return True

    def update(self):
        """compressai GaussianConditional.update: one quantised CDF per scale-table entry (float32 on the host, once)."""
        dev = self.scale_bound.device
        table = self.scale_table.detach().float().cpu()
        multiplier = -self._standardized_quantile(self.tail_mass / 2)
        pmf_center = torch.ceil(table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(torch.max(pmf_length))
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        samples_scale = table.unsqueeze(1)
        cum = lambda z: 0.5 * torch.erfc(-(2 ** -0.5) * z)
        upper = cum((0.5 - samples) / samples_scale)
        lower = cum((-0.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._offset = (-pmf_center).int().to(dev)
        self._cdf_length = (pmf_length + 2).int().to(dev)

    def build_indexes(self, scales):
        """compressai GaussianConditional.build_indexes: index of the first scale-table entry >= max(scale, bound)
        (= len(table)-1 minus the number of the first len-1 entries that are >= the bounded scale)."""
        s = torch.clamp(scales, min=float(self.scale_bound))           # lower_bound_scale (forward value)
        table = self.scale_table.to(s.device)
        # entries t with s <= t among table[:-1]  ==  (len-1) - #(t < s)  ->  index = #(table[:-1] < s)
        return torch.bucketize(s.contiguous(), table[:-1].contiguous(), right=False).int()

    def bits(self, inputs, params, noise=None, want_q=False, bit_sum=None):
        """Fused path: inputs (P,B,C,h,w), params (P,B,2C,h,w) -> (-log2 likelihood, quantised)."""
        return ops.gauss_rate(inputs, params, noise, want_q, bit_sum)

    def forward(self, inputs, scales, means=None, training=None):
        """API-compatible form: -> (outputs, likelihood).  4-D (B,C,h,w) tensors."""
        training = self.training if training is None else training
        B, C, h, w = inputs.shape
        mu = means if means is not None else torch.zeros_like(scales)
        params = torch.stack((scales, mu), 2).reshape(1, B, 2 * C, h, w).contiguous()
        noise = torch.empty_like(inputs).uniform_(-0.5, 0.5)[None].contiguous() if training else None
        bits, q = ops.gauss_rate(inputs[None].contiguous(), params, noise, want_q=True)
        return q[0], torch.exp2(-bits[0])


class EntropyBottleneck(_EntropyModelBase):
    """compressai.entropy_models.EntropyBottleneck(channels) with the default filters (3,3,3,3), init_scale 10."""

    def __init__(self, channels, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kw):
        super().__init__(**kw)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        if self.filters != (3, 3, 3, 3):
            raise ValueError("the HIP factorized-rate kernel is built for filters (3,3,3,3)")
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        f = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        C = self.channels
        for i in range(len(self.filters) + 1):
            init = float(np.log(np.expm1(1 / scale / f[i + 1])))
            self.register_parameter("_matrix%d" % i, nn.Parameter(torch.full((C, f[i + 1], f[i]), init)))
            self.register_parameter("_bias%d" % i, nn.Parameter(torch.empty(C, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter("_factor%d" % i, nn.Parameter(torch.zeros(C, f[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.Tensor([-self.init_scale, 0, self.init_scale]).repeat(C, 1, 1))
        target = float(np.log(2 / self.tail_mass - 1))
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    def packed(self):
        return pack_entropy_bottleneck({k: v.detach() for k, v in self.named_parameters()})

    def forward(self, x, training=None):
        """-> (outputs, likelihood) for a (B,C,h,w) tensor."""
        training = self.training if training is None else training
        xi = x[None].contiguous()
        noise = torch.empty_like(xi).uniform_(-0.5, 0.5) if training else None
        bits, q = ops.factorized_rate(xi, self.packed()[None].contiguous(), noise)
        return q[0], torch.exp2(-bits[0])

    # ---- real coding (compressai EntropyBottleneck.update / compress / decompress): the tables are a few hundred numbers per
    # channel, built on the host in float32 like compressai's; the symbols are produced on the GPU
    def _logits_cumulative(self, inputs):
        """compressai EntropyBottleneck._logits_cumulative (stop_gradient): inputs (C,1,N) -> logits (C,1,N)."""
        import torch.nn.functional as F
        logits = inputs
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(F.softplus(getattr(self, "_matrix%d" % i).detach()), logits)
            logits = logits + getattr(self, "_bias%d" % i).detach()
            if i < len(self.filters):
                logits = logits + torch.tanh(getattr(self, "_factor%d" % i).detach()) * torch.tanh(logits)
        return logits

    def _get_medians(self):
        return self.quantiles[:, :, 1:2].detach()                       # (C,1,1)

    def update(self, force=False):
        """compressai EntropyBottleneck.update: one quantised CDF per channel from the learned density, centred on the
        channel's median; offsets = -ceil(median - lower quantile)."""
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.quantiles.device
        with torch.no_grad():
            q = self.quantiles.detach()
            medians = q[:, 0, 1]
            minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
            maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
            pmf_start = medians - minima
            pmf_length = maxima + minima + 1
            max_length = int(pmf_length.max())
            samples = torch.arange(max_length, device=dev)[None, :] + pmf_start[:, None, None]      # (C,1,N)
            lower = self._logits_cumulative(samples - 0.5)
            upper = self._logits_cumulative(samples + 0.5)
            sign = -torch.sign(lower + upper)
            pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
            tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
            self._quantized_cdf = self._pmf_to_cdf(pmf.cpu(), tail_mass.cpu(), pmf_length.cpu(), max_length).to(dev)
            self._offset = (-minima).int().to(dev)
            self._cdf_length = (pmf_length + 2).int().to(dev)
        return True

    def symbols_and_indexes(self, x):
        """x (..., C, h, w) -> (symbols = round(x - median) int32, indexes = channel number int32), same shape
        (compressai compress: quantize(x, "symbols", medians), build_indexes)."""
        C = self.channels
        med = self._get_medians().reshape(C, 1, 1).to(x.device)
        sym = torch.round(x - med).int()
        idx = torch.arange(C, device=x.device, dtype=torch.int32).reshape(C, 1, 1).expand_as(x).contiguous()
        return sym, idx

    def dequantize_symbols(self, sym):
        """symbols (..., C, h, w) -> values = symbol + median (compressai decompress)."""
        return sym.float() + self._get_medians().reshape(self.channels, 1, 1).to(sym.device)

    def loss(self):
        """Auxiliary quantile loss |logits(quantiles) - target| (compressai EntropyBottleneck.loss): a 3-point host-side
        evaluation per channel, not on the per-batch hot path."""
        import torch.nn.functional as F
        logits = self.quantiles
        for i in range(5):
            logits = torch.matmul(F.softplus(getattr(self, "_matrix%d" % i).detach()), logits)
            logits = logits + getattr(self, "_bias%d" % i).detach()
            if i < 4:
                logits = logits + torch.tanh(getattr(self, "_factor%d" % i).detach()) * torch.tanh(logits)
        return torch.abs(logits - self.target).sum()
