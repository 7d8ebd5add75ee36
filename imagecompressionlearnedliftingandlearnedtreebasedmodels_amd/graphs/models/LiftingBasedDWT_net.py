"""Model wrapper + entropy layers on the HIP kernels -- mirrors graphs/models/LiftingBasedDWT_net.py of the reference:
``LiftingBasedDWTNetWrapper`` (:35-99), ``LiftingBasedDWTNet`` (:100-180), ``DWTFactorizedEntropyLayer`` (:182-231),
``DWTConditioned2EntropyLayerZTsepSubbands`` (:233-372), ``onlyEZWT`` (:759-840).

Same class names, constructor signature (one ``config`` object), dispatch strings, forward signature and state_dict
layout.  The three per-plane networks of ``clrch == 1`` are executed TOGETHER: every kernel launch covers the three
planes and the whole batch (plane-major tensors, per-plane weights stacked), instead of three sequential sub-networks.

Real entropy coding (``compress`` / ``test`` / ``compress_ar`` / ``decompress_ar``, :76-99,136-152,374-556) is built for
the layer the reference builds it for, ``DWTConditioned2EntropyLayerZTsepSubbands``: the per-pixel Python loops become a
wavefront schedule on the GPU (entropy_coding.py) and the range coder is the C-ABI's host rANS (ans.py).
"""
import math

import os

import torch
import torch.nn.functional as F
from torch import nn

from ... import autograd as ag
from ... import ops
from ...entropy_models import EntropyBottleneck, GaussianConditional
from ...packed_cache import PackedOwnerMixin, cached
from ... import param_arena
from ..layers.lifting_dwt_nets import (DWTPytorchWaveletsLayer, LiftingBasedNeuralWaveletv4, _stack,
                                       decode_planes, encode_planes, lifting_coeff)
from ..layers.masked_conv2d import MaskedConv2d

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):
    """LiftingBasedDWT_net.py:32-33: 64 log-spaced Gaussian scales from 0.11 to 256."""
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


def byte_extractor(string):
    """LiftingBasedDWT_net.py:15-17 (bytes of one stream; a list of per-image streams counts all of them)."""
    return len(string) if isinstance(string, (bytes, bytearray)) else sum(len(s) for s in string)


def byte_extractor_xo(strings):
    """LiftingBasedDWT_net.py:19-31 for this layer's return value: one stream (or per-image list) per level."""
    return sum(byte_extractor(s) for s in strings)


# ------------------------------------------------------------------------------------------------ helpers
def _conv_params(mods, tag):
    """Stacked (weight, bias) of the same conv layer of every plane (cached until a parameter changes)."""
    for m in mods:
        if isinstance(m, MaskedConv2d):
            m.apply_mask_()          # reference: weight.data *= mask on every forward (masked_conv2d.py:20)
    m0 = mods[0]
    mask = m0.tap_bits() if isinstance(m0, MaskedConv2d) else None

    def build():
        w = _stack(mods, lambda m: m.weight)
        return w, _stack(mods, lambda m: m.bias), ops.conv_pack(w, m0.kernel_size[0], m0.groups, tap_mask=mask), mask
    return cached(m0, ("conv", tag), [p for m in mods for p in (m.weight, m.bias)], build)


def _conv(mods, x, act=ops.ACT_NONE, upsample2=False, **kw):
    m = mods[0]
    w, b, packed, mask = _conv_params(mods, "w")
    return ops.conv2d(x, w, b, m.kernel_size[0], groups=m.groups, act=act, upsample2=upsample2, tap_mask=mask,
                      packed=packed, **kw)


def _plc_pair(first, second, parent, act2):
    """The tree-context pair conv3x3(3 -> 243) on the 2x-upsampled parent, LeakyReLU, conv3x3(243 -> 243)
    (LiftingBasedDWT_net.py:271-272,348,355 / :793-795,822).  The second, dense conv is 61 % of the headline step's
    FLOPs: in mode 'f16x3' (ops.plc_mode) it runs on the fp16 matrix cores with split-fp16 operands (fp32-level
    accuracy, csrc/conv_f16x3.hip); its activation scale comes from the first conv's epilogue (absmax slots)."""
    if ops.plc_mode() != "f16x3":
        t = _conv(first, parent, act=ops.ACT_LRELU, upsample2=True)
        return _conv(second, t, act=act2)
    m2 = second[0]
    if ops.storage_dtype() == "fp16":
        # fp16 STORAGE of the 243-channel tensor between the two convs (BASELINE configs[4]): power-of-two scale from a
        # bound on |LeakyReLU(conv1(parent))| <= max|parent| * max row L1 norm + max|bias| (all on the device, no sync)
        w1, b1, packed1, _ = _conv_params(first, "w")
        l1, bm = cached(first[0], ("l1bound",), [p for m in first for p in (m.weight, m.bias)],
                        lambda: (w1.abs().sum(dim=(2, 3, 4)).amax(dim=1), b1.abs().amax(dim=1)))
        bound = ops.absmax_slots(parent).amax(dim=1) * l1 + bm
        oscale = torch.exp2(14.0 - torch.ceil(torch.log2(bound.clamp_min(1e-30)))).contiguous()
        t16 = ops.conv2d_f16out(parent, w1, b1, 3, oscale, act=ops.ACT_LRELU, upsample2=True, packed=packed1)
        b2, packed16 = cached(m2, ("conv_f16x3",), [p for m in second for p in (m.weight, m.bias)],
                              lambda: (_stack(second, lambda m: m.bias), ops.conv_f16x3_pack(_stack(second, lambda m: m.weight))))
        return ops.conv3x3_f16in(t16, packed16, b2, m2.out_channels, oscale, act=act2)
    def build():
        return _stack(second, lambda m: m.bias), ops.conv_f16x3_pack(_stack(second, lambda m: m.weight))
    b2, packed16 = cached(m2, ("conv_f16x3",), [p for m in second for p in (m.weight, m.bias)], build)
    m1 = first[0]
    if ops.plc_fuse() and m1.in_channels == 3 and m1.kernel_size[0] == 3 and m1.out_channels <= 256 and m1.groups == 1:
        # one launch: the first conv is computed per halo patch inside the second's staging (no 243-channel tensor in HBM)
        packed1f = cached(m1, ("plc_fused1",), [p for m in first for p in (m.weight, m.bias)],
                          lambda: ops.plc_fused_pack1(_stack(first, lambda m: m.weight), _stack(first, lambda m: m.bias)))
        return ops.plc_fused(parent, packed1f, packed16, b2, m1.out_channels, m2.out_channels, act=act2)
    slots = torch.empty(parent.shape[0], 64, device=parent.device, dtype=torch.float32)
    t = _conv(first, parent, act=ops.ACT_LRELU, upsample2=True, absmax=slots)
    return ops.conv3x3_f16x3(t, packed16, b2, m2.out_channels, act=act2, slots=slots)


def _noise(t, training):
    return torch.empty_like(t).uniform_(-0.5, 0.5) if training else None


def _seq_stack(seqs, x, idxs):
    """Masked-conv stack with LeakyReLU between layers (LiftingBasedDWT_net.py:299-305,311-317)."""
    t = x
    for n in idxs:
        t = _conv([s[n] for s in seqs], t, act=ops.ACT_NONE if n == idxs[-1] else ops.ACT_LRELU)
    return t


def _eb_packed(ebs):
    return cached(ebs[0], ("eb",), [p for e in ebs for p in e.parameters()],
                      lambda: torch.stack([e.packed() for e in ebs], 0).contiguous())


# ------------------------------------------------------------------------------------------------ entropy layers
class _EntropyLayerBase(PackedOwnerMixin, nn.Module):
    def _level_channels(self, config):
        self.num_lifting_layers = config.dwtlevels
        assert self.num_lifting_layers > 0
        self.clrch = config.clrch
        self.se, self.so = 1, 3
        self.ses = [self.se * self.clrch] * self.num_lifting_layers
        self.sos = [self.so * self.clrch] * self.num_lifting_layers

    def forward(self, out_xe, out_xo_list):
        si_xe, si_xo, xe_q, xo_q = self.forward_planes([self], out_xe[None].contiguous(),
                                                       [t[None].contiguous() for t in out_xo_list], self.training)
        return si_xe[0], [t[0] for t in si_xo], xe_q[0], [t[0] for t in xo_q]


class DWTFactorizedEntropyLayer(_EntropyLayerBase):
    """Factorized model, one EntropyBottleneck per level (LiftingBasedDWT_net.py:182-231)."""

    def __init__(self, config):
        super().__init__()
        self._init_packed_owner()
        self._level_channels(config)
        self.ent_out_xo_list = nn.ModuleList()
        self.scl_out_xo_list = nn.ParameterList()     # present in the reference's state_dict, unused in forward (:205-211)
        self.scb_out_xo_list = nn.ParameterList()
        for i in range(self.num_lifting_layers):
            self.ent_out_xo_list.append(EntropyBottleneck(channels=self.sos[i]))
            self.scl_out_xo_list.append(nn.Parameter(torch.full((1, self.sos[i], 1, 1), i + 1.0)))
            self.scb_out_xo_list.append(nn.Parameter(torch.full((1, self.sos[i], 1, 1), 1.0)))
        se = self.ses[-1]
        self.ent_out_xe = EntropyBottleneck(channels=se)
        self.scl_out_xe = nn.Parameter(torch.full((1, se, 1, 1), 5.0))
        self.scb_out_xe = nn.Parameter(torch.full((1, se, 1, 1), 1.0 / 5.0))

    @staticmethod
    def forward_planes(layers, out_xe, out_xo_list, training):
        si_xo, q_xo = [], []
        for i in range(len(out_xo_list)):
            ebs = [l.ent_out_xo_list[i] for l in layers]
            bits, q = ops.factorized_rate(out_xo_list[i], _eb_packed(ebs), _noise(out_xo_list[i], training))
            si_xo.append(bits)
            q_xo.append(q)
        bits, q = ops.factorized_rate(out_xe, _eb_packed([l.ent_out_xe for l in layers]), _noise(out_xe, training))
        return bits, si_xo, q, q_xo


def _plc(so):
    o = so * 81
    return nn.Conv2d(so, o, kernel_size=3, stride=1, padding=1), nn.LeakyReLU(), nn.Conv2d(o, o, kernel_size=3, stride=1, padding=1)


class onlyEZWT(_EntropyLayerBase):
    """Inter-subband ("zero-tree") model only (LiftingBasedDWT_net.py:759-840)."""

    def __init__(self, config):
        super().__init__()
        self._init_packed_owner()
        self._level_channels(config)
        self.config = config
        self.plc_list = nn.ModuleList()
        self.ent_out_xo_list = nn.ModuleList()
        for i in range(self.num_lifting_layers - 1):
            a, r, b = _plc(self.sos[i + 1])
            self.plc_list.append(nn.Sequential(a, r, b, nn.LeakyReLU(), nn.Conv2d(self.sos[i + 1] * 81, 6, kernel_size=1)))
            self.ent_out_xo_list.append(GaussianConditional(scale_table=None, scale_bound=0.11))
        self.ent_out_xe = EntropyBottleneck(channels=1)
        self.ent_out_xo = EntropyBottleneck(channels=3)

    @staticmethod
    def forward_planes(layers, out_xe, out_xo_list, training):
        L = len(out_xo_list)
        si_list, q_list = [], []
        si_xe, xe_q = ops.factorized_rate(out_xe, _eb_packed([l.ent_out_xe for l in layers]), _noise(out_xe, training))
        bits, q = ops.factorized_rate(out_xo_list[L - 1], _eb_packed([l.ent_out_xo for l in layers]),
                                      _noise(out_xo_list[L - 1], training))
        si_list.append(bits)
        q_list.append(q)
        parent = q
        for i in range(L - 2, -1, -1):
            seqs = [l.plc_list[i] for l in layers]
            t = _plc_pair([s[0] for s in seqs], [s[2] for s in seqs], parent, ops.ACT_LRELU)   # :822,835 + :793-794
            ms = _conv([s[4] for s in seqs], t)
            bits, q = ops.gauss_rate(out_xo_list[i], ms, _noise(out_xo_list[i], training), want_q=True)   # :832
            si_list.append(bits)
            q_list.append(q)
            parent = q
        q_list.reverse()
        si_list.reverse()
        return si_xe, si_list, xe_q, q_list

    # ---------------------------------------------------------------- real entropy coding (EXTENSION: the reference's
    # onlyEZWT defines only forward, :759-840; this is what its compress would be with compressai's own
    # EntropyBottleneck.compress / GaussianConditional.compress).  Every level's (sigma, mu) depend only on the decoded parent
    # level, so each tensor is coded in ONE parallel pass (no wavefront): xe and the coarsest level with the factorized
    # priors, the finer levels with the Gaussian tables of get_scale_table().  One rANS stream per (plane, image, tensor).
    @staticmethod
    def _level_params(layers, i, parent):
        seqs = [l.plc_list[i] for l in layers]
        t = _plc_pair([s[0] for s in seqs], [s[2] for s in seqs], parent, ops.ACT_LRELU)
        return _conv([s[4] for s in seqs], t)                                       # (P,B,6,h,w): sigma even, mu odd

    @staticmethod
    def compress_planes(layers, out_xe, out_xo_list):
        """-> (strings_xe[p][b], [strings_xo[p][b]] finest first, xe_q, [xo_q] finest first)."""
        from . import entropy_coding as ec
        L = len(out_xo_list)
        with torch.no_grad():
            s_xe, xe_q = ec.code_factorized([l.ent_out_xe for l in layers], out_xe, out_xe.shape)
            s, q = ec.code_factorized([l.ent_out_xo for l in layers], out_xo_list[L - 1], out_xo_list[L - 1].shape)
            s_list, q_list = [s], [q]
            for i in range(L - 2, -1, -1):
                tabs = ec._Tables(layers[0].ent_out_xo_list[i], get_scale_table())
                for l in layers[1:]:
                    l.ent_out_xo_list[i].update_scale_table(get_scale_table())
                ms = onlyEZWT._level_params(layers, i, q)
                s, q = ec.code_gaussian_parallel([l.ent_out_xo_list[i] for l in layers], ms, out_xo_list[i],
                                                 out_xo_list[i].shape, tabs)
                s_list.append(s)
                q_list.append(q)
        s_list.reverse()
        q_list.reverse()
        return s_xe, s_list, xe_q, q_list

    @staticmethod
    def decompress_planes(layers, strings_xe, strings_xo_list, shape_xe, shapes_xo):
        """strings -> (xe, [xo] finest first), bit-identical to compress_planes' dequantised tensors."""
        from . import entropy_coding as ec
        L = len(shapes_xo)
        with torch.no_grad():
            _, xe = ec.code_factorized([l.ent_out_xe for l in layers], None, shape_xe, strings_xe)
            _, q = ec.code_factorized([l.ent_out_xo for l in layers], None, shapes_xo[L - 1], strings_xo_list[L - 1])
            q_list = [q]
            for i in range(L - 2, -1, -1):
                tabs = ec._Tables(layers[0].ent_out_xo_list[i], get_scale_table())
                for l in layers[1:]:
                    l.ent_out_xo_list[i].update_scale_table(get_scale_table())
                ms = onlyEZWT._level_params(layers, i, q)
                _, q = ec.code_gaussian_parallel([l.ent_out_xo_list[i] for l in layers], ms, None, shapes_xo[i], tabs,
                                                 strings_xo_list[i])
                q_list.append(q)
        q_list.reverse()
        return xe, q_list

    @staticmethod
    def test_planes(layers, out_xe, out_xo_list):
        """Encode, then decode FROM THE STRINGS (same contract as the conditioned2 layer's test_planes)."""
        s_xe, s_xo, _, _ = onlyEZWT.compress_planes(layers, out_xe, out_xo_list)
        xe, xo = onlyEZWT.decompress_planes(layers, s_xe, s_xo, out_xe.shape, [t.shape for t in out_xo_list])
        return s_xe, s_xo, xe, xo

    def test(self, out_xe, out_xo_list):
        s_xe, s_xo, xe, xo = self.test_planes([self], out_xe[None].contiguous(), [t[None].contiguous() for t in out_xo_list])
        one = lambda rows: rows[0][0] if len(rows[0]) == 1 else rows[0]
        return one(s_xe), [one(r) for r in s_xo], xe[0], [t[0] for t in xo]


class DWTConditioned2EntropyLayerZTsepSubbands(_EntropyLayerBase):
    """Tree (parent level) + causal intra-subband context model (LiftingBasedDWT_net.py:233-372)."""

    def __init__(self, config):
        super().__init__()
        self._init_packed_owner()
        self._level_channels(config)
        self.config = config
        self.scale_table = get_scale_table()              # plain attribute, as in the reference (:245)
        L = self.num_lifting_layers
        self.plc_list = nn.ModuleList()
        self.csc_list = nn.ModuleList()
        self.cgp_out_xo_list = nn.ModuleList()
        self.ent_out_xo_list = nn.ModuleList()
        self.scl_out_xo_list = nn.ParameterList()
        self.scb_out_xo_list = nn.ParameterList()
        for i in range(L - 1):
            inn1 = self.sos[i + 1]
            self.plc_list.append(nn.Sequential(*_plc(inn1)))
            inn2 = self.sos[i]
            self.csc_list.append(MaskedConv2d("A", inn2, inn2 * 81, 5, 1, 2, groups=inn2))
            inn = inn1 * 81 + inn2 * 81
            self.cgp_out_xo_list.append(nn.Sequential(
                nn.Conv2d(inn, inn, 1, groups=inn1), nn.LeakyReLU(inplace=True),
                nn.Conv2d(inn, inn // 3, 1, groups=inn1), nn.LeakyReLU(inplace=True),
                nn.Conv2d(inn // 3, inn // 9, 1, groups=inn1), nn.LeakyReLU(inplace=True),
                nn.Conv2d(inn // 9, self.sos[i] * 2, 1, groups=inn1)))
            self.ent_out_xo_list.append(GaussianConditional(scale_table=None, scale_bound=0.11))

        def stack(g):
            o = g * 81
            spec = [("A", g, o), ("B", o, o), ("B", o, o // 3), ("B", o // 3, o // 9), ("B", o // 9, g * 2)]
            mods = []
            for n, (t, a, b) in enumerate(spec):
                mods.append(MaskedConv2d(t, a, b, 3, 1, 1, groups=g))
                if n != 4:
                    mods.append(nn.LeakyReLU(inplace=True))
            return nn.Sequential(*mods)
        self.csc_list.append(stack(self.sos[L - 1]))
        self.ent_out_xo_list.append(GaussianConditional(scale_table=None, scale_bound=0.11))
        self.csc_xe = stack(self.ses[L - 1])
        self.ent_out_xe = GaussianConditional(scale_table=None, scale_bound=0.11)

    @staticmethod
    def forward_planes(layers, out_xe, out_xo_list, training):
        L = len(out_xo_list)
        idx5 = (0, 2, 4, 6, 8)
        # xe: decoder / context see quantize(x) WITHOUT means (:330); the rate uses an independent noise sample (:334)
        xe_q = ops.quantize(out_xe, _noise(out_xe, training))
        ms = _seq_stack([l.csc_xe for l in layers], xe_q, idx5)
        si_xe, _ = ops.gauss_rate(out_xe, ms, _noise(out_xe, training))
        q_list, si_list = [], []
        i = L - 1
        xo_q = ops.quantize(out_xo_list[i], _noise(out_xo_list[i], training))
        ms = _seq_stack([l.csc_list[i] for l in layers], xo_q, idx5)
        bits, _ = ops.gauss_rate(out_xo_list[i], ms, _noise(out_xo_list[i], training))
        si_list.append(bits)
        q_list.append(xo_q)
        parent = xo_q
        for i in range(L - 2, -1, -1):
            x = out_xo_list[i]
            P, B, so, h, w = x.shape
            xo_q = ops.quantize(x, _noise(x, training))
            seqs = [l.plc_list[i] for l in layers]
            plc = _plc_pair([s[0] for s in seqs], [s[2] for s in seqs], parent, ops.ACT_NONE)                # :348,355
            cg = [l.cgp_out_xo_list[i] for l in layers]
            cs = [l.csc_list[i] for l in layers]
            convs = [[s[n] for s in cg] for n in (0, 2, 4, 6)]
            for m in cs:
                m.apply_mask_()
            # The masked csc conv (:275-277,353) feeds cgp layer 0 (:282) with no nonlinearity in between and the regroup
            # (:357-359) is pure data movement, so its 81 channels per subband are folded into layer 0:
            #   W0[:, csc half] . Wcsc -> 12 extra input columns = the live taps of the quantised subband itself,
            # gathered inside the fused kernel.  The csc conv, its 243-channel output and half of layer 0's MACs disappear.
            packed, dims, packed16 = cached(cg[0], ("cgp_ctx",),
                                            [p for layer in convs for m in layer for p in (m.weight, m.bias)] +
                                            [p for m in cs for p in (m.weight, m.bias)],
                                            lambda: _fold_csc_into_cgp(convs, cs, so))
            if packed16 is not None and not training and ops.cgp_mode() == "f16x3":
                # split-fp16 register chain -> (sigma, mu), then the fused Gaussian rate kernel
                params = ops.cgp16_params(plc, xo_q, packed16, cs[0].kernel_size[0], cs[0].tap_bits())
                bits, _ = ops.gauss_rate(x, params)
            else:
                bits = ops.cgp_rate_ctx(plc, xo_q, x, packed, dims, cs[0].kernel_size[0], cs[0].tap_bits(), _noise(x, training))
            si_list.append(bits)
            q_list.append(xo_q)
            parent = xo_q
        q_list.reverse()
        si_list.reverse()
        return si_xe, si_list, xe_q, q_list


    # ---------------------------------------------------------------- real entropy coding (:374-556)
    def test(self, out_xe, out_xo_list):
        """Reference signature (:374): (B,C,h,w) tensors of ONE plane -> (string_xe, [string_xo per level, finest first],
        decoded xe, [decoded xo]).  Strings are ``bytes`` at batch 1 (as in the reference), lists of per-image bytes else."""
        s_xe, s_xo, xe, xo = self.test_planes([self], out_xe[None].contiguous(), [t[None].contiguous() for t in out_xo_list])
        one = lambda rows: rows[0][0] if len(rows[0]) == 1 else rows[0]
        return one(s_xe), [one(r) for r in s_xo], xe[0], [t[0] for t in xo]

    @staticmethod
    def test_planes(layers, out_xe, out_xo_list):
        """Encode, then decode FROM THE STRINGS (the reference returns the decoder's tensors, :419-456)."""
        cls = DWTConditioned2EntropyLayerZTsepSubbands
        s_xe, s_xo, _, _ = cls.compress_planes(layers, out_xe, out_xo_list)
        xe, xo = cls.decompress_planes(layers, s_xe, s_xo, out_xe.shape, [t.shape for t in out_xo_list])
        return s_xe, s_xo, xe, xo

    @staticmethod
    def _coding_setup(layers):
        from .entropy_coding import _Tables
        l0 = layers[0]
        tabs = l0.__dict__.get("_rans_tables")
        if tabs is None:
            tabs = l0.__dict__["_rans_tables"] = _Tables(l0.ent_out_xe, l0.scale_table)
        for l in layers:                                   # every Gaussian model carries the same table (:462)
            for em in [l.ent_out_xe] + list(l.ent_out_xo_list):
                em.update_scale_table(l.scale_table)
        stack = lambda seqs, crop: _seq_stack(seqs, crop, (0, 2, 4, 6, 8))
        return tabs, stack

    @staticmethod
    def _tree_context(layers, i, parent, so):
        seqs = [l.plc_list[i] for l in layers]
        plc = _plc_pair([s[0] for s in seqs], [s[2] for s in seqs], parent, ops.ACT_NONE)
        cg = [l.cgp_out_xo_list[i] for l in layers]
        cs = [l.csc_list[i] for l in layers]
        convs = [[s[n] for s in cg] for n in (0, 2, 4, 6)]
        for m in cs:
            m.apply_mask_()
        packed, dims, packed16 = cached(cg[0], ("cgp_ctx",), [p for layer in convs for m in layer for p in (m.weight, m.bias)] +
                                        [p for m in cs for p in (m.weight, m.bias)], lambda: _fold_csc_into_cgp(convs, cs, so))
        return plc, (packed, packed16), dims, cs[0].kernel_size[0], cs[0].tap_bits()

    @staticmethod
    def compress_planes(layers, out_xe, out_xo_list):
        """compress_ar for every tensor (:386-417): -> (strings_xe[p][b], [strings_xo[p][b]] finest first, xe_q, [xo_q])
        with *_q = round(y - mu) + mu, the values the decoder reconstructs."""
        from . import entropy_coding as ec
        tabs, stack = DWTConditioned2EntropyLayerZTsepSubbands._coding_setup(layers)
        L = len(out_xo_list)
        with torch.no_grad():
            s_xe, xe_q = ec.code_crop_stack(stack, [l.ent_out_xe for l in layers], [l.csc_xe for l in layers], out_xe,
                                            out_xe.shape, tabs)
            s, q = ec.code_crop_stack(stack, [l.ent_out_xo_list[L - 1] for l in layers], [l.csc_list[L - 1] for l in layers],
                                      out_xo_list[L - 1], out_xo_list[L - 1].shape, tabs)
            s_list, q_list = [s], [q]
            for i in range(L - 2, -1, -1):
                x = out_xo_list[i]
                plc, packed, dims, K, bits = DWTConditioned2EntropyLayerZTsepSubbands._tree_context(layers, i, q, x.shape[2])
                em_i = [l.ent_out_xo_list[i] for l in layers]
                if packed[1] is not None:       # the reference's cgp widths: one fused launch per wavefront step
                    s, q = ec.code_tree_level(em_i, plc, packed[1], K, bits, x, x.shape, tabs)
                else:
                    s, q = ec.code_tree_level_generic(em_i, plc, packed[0], dims, K, bits, x, x.shape, tabs)
                s_list.append(s)
                q_list.append(q)
        s_list.reverse()
        q_list.reverse()
        return s_xe, s_list, xe_q, q_list

    @staticmethod
    def decompress_planes(layers, strings_xe, strings_xo_list, shape_xe, shapes_xo):
        """decompress_ar for every tensor (:419-454): strings -> (xe, [xo] finest first), bit-identical to compress_planes'
        dequantised tensors."""
        from . import entropy_coding as ec
        tabs, stack = DWTConditioned2EntropyLayerZTsepSubbands._coding_setup(layers)
        L = len(shapes_xo)
        with torch.no_grad():
            _, xe = ec.code_crop_stack(stack, [l.ent_out_xe for l in layers], [l.csc_xe for l in layers], None, shape_xe, tabs,
                                       strings_xe)
            _, q = ec.code_crop_stack(stack, [l.ent_out_xo_list[L - 1] for l in layers], [l.csc_list[L - 1] for l in layers],
                                      None, shapes_xo[L - 1], tabs, strings_xo_list[L - 1])
            q_list = [q]
            for i in range(L - 2, -1, -1):
                plc, packed, dims, K, bits = DWTConditioned2EntropyLayerZTsepSubbands._tree_context(layers, i, q, shapes_xo[i][2])
                em_i = [l.ent_out_xo_list[i] for l in layers]
                if packed[1] is not None:
                    _, q = ec.code_tree_level(em_i, plc, packed[1], K, bits, None, shapes_xo[i], tabs, strings_xo_list[i])
                else:
                    _, q = ec.code_tree_level_generic(em_i, plc, packed[0], dims, K, bits, None, shapes_xo[i], tabs,
                                                      strings_xo_list[i])
                q_list.append(q)
        q_list.reverse()
        return xe, q_list


class DWTConditioned2EntropyLayerZTBlock(_EntropyLayerBase):
    """Block-wise variant (LiftingBasedDWT_net.py:558-757): the four polyphase phases of each subband are predicted in
    sequence from the (not upsampled) parent subband and the phases already coded, by eight 5-layer CNNs (mu / sigma)."""

    def __init__(self, config):
        super().__init__()
        self._init_packed_owner()
        self._level_channels(config)
        self.dwtLevels = config.dwtlevels
        self.multiplier = 8
        hid = 32
        for k in range(1, 5):
            setattr(self, "dep_%d_list_mu" % k, nn.ModuleList())
        for k in range(1, 5):
            setattr(self, "dep_%d_list_sigma" % k, nn.ModuleList())
        self.cgp_out_xo_list = nn.ModuleList()
        self.ent_out_xo_list = nn.ModuleList()
        self.scl_out_xo_list = nn.ParameterList()
        self.scb_out_xo_list = nn.ParameterList()

        def net(cin):
            return nn.Sequential(nn.Conv2d(cin, hid, 3, padding=1), nn.LeakyReLU(inplace=True),
                                 nn.Conv2d(hid, hid, 3, padding=1), nn.LeakyReLU(inplace=True),
                                 nn.Conv2d(hid, hid, 1), nn.LeakyReLU(inplace=True),
                                 nn.Conv2d(hid, hid, 1), nn.LeakyReLU(inplace=True), nn.Conv2d(hid, 1, 1))
        for i in range(self.num_lifting_layers - 1):
            for _ in range(3):
                self.ent_out_xo_list.append(GaussianConditional(scale_table=None, scale_bound=0.11))
                self.scl_out_xo_list.append(nn.Parameter(torch.full((1, self.sos[i], 1, 1), i * 1.0 + 1.0)))
                self.scb_out_xo_list.append(nn.Parameter(torch.full((1, self.sos[i], 1, 1), 1.0)))
                for k in range(1, 5):
                    getattr(self, "dep_%d_list_mu" % k).append(net(k))
                for k in range(1, 5):
                    getattr(self, "dep_%d_list_sigma" % k).append(net(k))
        self.ent_out_xo_list.append(GaussianConditional(scale_table=None, scale_bound=0.11))
        self.gaussian_conditional = GaussianConditional(None)
        self.ent_out_xe = EntropyBottleneck(channels=1)
        self.ent_out_xo = EntropyBottleneck(channels=3)

    @staticmethod
    def forward_planes(layers, out_xe, out_xo_list, training):
        L = len(out_xo_list)
        si_xe, xe_q = ops.factorized_rate(out_xe, _eb_packed([l.ent_out_xe for l in layers]), _noise(out_xe, training))
        bits, q = ops.factorized_rate(out_xo_list[L - 1], _eb_packed([l.ent_out_xo for l in layers]),
                                      _noise(out_xo_list[L - 1], training))
        si_list, q_list = [bits], [q]
        con = q
        slots = ((0, 0), (0, 1), (1, 0), (1, 1))
        for i in range(L - 1):
            lev = L - i - 2
            x = out_xo_list[lev]
            P, B, _, H, W = x.shape
            sis, qs = [], []
            for j in range(3):
                xj = x[:, :, j:j + 1].contiguous()
                qj = ops.quantize(xj, _noise(xj, training))                                   # :716-718
                ee, eo, oe = qj[..., 0::2, 0::2], qj[..., 0::2, 1::2], qj[..., 1::2, 0::2]    # :719-721
                dep1 = con[:, :, j:j + 1]
                deps = (dep1, torch.cat((dep1, ee), 2), torch.cat((dep1, ee, eo), 2), torch.cat((dep1, ee, eo, oe), 2))
                params = torch.empty(P, B, 2, H, W, device=x.device, dtype=torch.float32)     # (sigma, mu)
                idx = j + i * 3
                for k in range(4):                                                            # :723-740
                    d = deps[k].contiguous()
                    for ch, kind in ((1, "mu"), (0, "sigma")):
                        seqs = [getattr(l, "dep_%d_list_%s" % (k + 1, kind))[idx] for l in layers]
                        t = d
                        for n in (0, 2, 4, 6, 8):
                            t = _conv([s_[n] for s_ in seqs], t, ops.ACT_NONE if n == 8 else ops.ACT_LRELU)
                        params[:, :, ch:ch + 1, slots[k][0]::2, slots[k][1]::2] = t          # strided placement
                bits, _ = ops.gauss_rate(xj, params, _noise(xj, training))                    # :743-744
                sis.append(bits)
                qs.append(qj)
            si_list.append(torch.cat(sis, 2))
            con = torch.cat(qs, 2)
            q_list.append(con)
        q_list.reverse()
        si_list.reverse()
        return si_xe, si_list, xe_q, q_list


# ------------------------------------------------------------------------------------------------ training path
# Same maths as the eval path, but every op is a differentiable autograd.Function (forward AND backward are HIP
# kernels; torch only keeps the tape and un-stacks the per-plane parameter gradients).  Built for the headline
# configuration: LiftingBasedNeuralWaveletv4 + SubbandAutoEncoder + conditioned2ZTsepSubbands.
# LLDWT_CGP_TRAIN=cat keeps the concatenated [plc_g | taps_g] input of the training cgp stack (CgpRateFn instead of CgpRateCtxFn)
_CGP_TRAIN_CTX = os.environ.get("LLDWT_CGP_TRAIN", "ctx") != "cat"


def _tstack(mods, get):
    ts = [get(m) for m in mods]
    if all(isinstance(t, torch.nn.Parameter) for t in ts):
        return param_arena.stack_leaf(ts)         # a slice of the parameter arena when the group is laid out (param_arena.py)
    return torch.stack(ts, 0)


def _tconv(mods, x, act=ops.ACT_NONE, upsample2=False):
    m0 = mods[0]
    mask = None
    if isinstance(m0, MaskedConv2d):
        for m in mods:
            m.apply_mask_()
        mask = m0.tap_bits()
    return ag.conv(x, _tstack(mods, lambda m: m.weight), _tstack(mods, lambda m: m.bias), m0.kernel_size[0],
                   groups=m0.groups, act=act, upsample2=upsample2, tap_mask=mask)


def _tconv_transpose1x1(mods, x, act):
    """Grouped 1x1 ConvTranspose2d (SubbandAutoEncoder.ae_up) as the equivalent grouped 1x1 conv."""
    m0 = mods[0]
    G = m0.groups
    cin_g, cout_g = m0.in_channels // G, m0.out_channels // G
    w = _tstack(mods, lambda m: m.weight.view(G, cin_g, cout_g).transpose(1, 2).reshape(G * cout_g, cin_g, 1, 1))
    return ag.conv(x, w.contiguous(), _tstack(mods, lambda m: m.bias), 1, groups=G, act=act)


def _tconv_transpose3x3(mods, x):
    """Dense 3x3 ConvTranspose2d (stride 1, padding 1) as the equivalent conv: swap in/out, flip the taps."""
    w = _tstack(mods, lambda m: m.weight.transpose(0, 1).flip(-1, -2))
    return ag.conv(x, w.contiguous(), _tstack(mods, lambda m: m.bias), 3)


def _ae_train(aes, x, decode):
    from ..layers.lifting_dwt_nets import SubbandAutoEncoder
    seqs = [(a.ae_up if decode else a.ae_down) for a in aes]
    t = x
    if not isinstance(aes[0], SubbandAutoEncoder):           # SubbandAutoEncoderBerk: 3x3 convs + GDN (:139-150)
        for n in (0, 2, 4, 6):
            layer = [s_[n] for s_ in seqs]
            t = _tconv_transpose3x3(layer, t) if decode else _tconv(layer, t)
            if n != 6:
                g = [s_[n + 1] for s_ in seqs]
                t = ag.gdn_train(t, _tstack(g, lambda m: m.beta), _tstack(g, lambda m: m.gamma), decode, g[0].beta_min)
        return t
    if seqs[0][2].in_channels // seqs[0][2].groups == 32:
        # the reference's H = 32 (lifting_dwt_nets.py:98): one fused forward and one fused backward-data kernel
        wb = []
        for n in (0, 2, 4, 6):
            layer = [s_[n] for s_ in seqs]
            if decode:      # ConvTranspose2d (in, out/groups) -> the equivalent grouped Conv2d weight
                G = layer[0].groups
                cin_g, cout_g = layer[0].in_channels // G, layer[0].out_channels // G
                wb.append(_tstack(layer, lambda m: m.weight.view(G, cin_g, cout_g).transpose(1, 2)
                                  .reshape(G * cout_g, cin_g, 1, 1)).contiguous())
            else:
                wb.append(_tstack(layer, lambda m: m.weight))
            wb.append(_tstack(layer, lambda m: m.bias))
        return ag.SubbandMlpFn.apply(t.contiguous(), *wb)
    for n in (0, 2, 4, 6):
        act = ops.ACT_NONE if n == 6 else ops.ACT_TANH
        layer = [s_[n] for s_ in seqs]
        t = _tconv_transpose1x1(layer, t, act) if decode else _tconv(layer, t, act)
    return t


def _lift_params(nets):
    """taps (4,P,3) and the 8 stacked P/U tensors (nblocks,2,P,...) WITH autograd history to the module parameters."""
    nb = len(nets[0].P_blocks)
    # one stack (one copy kernel) per stacked tensor, viewed to its nested shape -- not a stack of stacks of stacks
    P = len(nets)
    taps = param_arena.stack_leaf([n.preProcessingList[j].weight for j in range(4) for n in nets]).view(4, P, 3)
    Wt = []
    for cn in ("conv1", "conv2", "conv3", "conv4"):
        for attr in ("weight", "bias"):
            flat = [getattr(getattr(getattr(n, kind)[b], cn), attr) for b in range(nb) for kind in ("P_blocks", "U_blocks")
                    for n in nets]
            Wt.append(param_arena.stack_leaf(flat).view(nb, 2, P, *flat[0].shape))
    n0 = nets[0]
    meta = dict(levels=n0.waveletLevel, C=n0.depth_scale, K=n0.conv_filter_size, rw=n0.res_connection_weight,
                linear=n0.linearityflag != 1, different=n0.blockprop != "same")
    nh = nl = None
    if n0.config.scale == 1:                              # wavelet_forward_v2.py:76-80: gains 0.8698.. + 0.1 nh, 1.1496.. + 0.1 nl
        nh = _tstack(nets, lambda n: lifting_coeff[4] + n.nh.reshape(()) * 0.1).contiguous()
        nl = _tstack(nets, lambda n: lifting_coeff[5] + n.nl.reshape(()) * 0.1).contiguous()
    return taps.contiguous(), meta, Wt, nh, nl


def _eb_train(ebs):
    """(P,C,59) packed EntropyBottleneck parameters WITH autograd history."""
    from ...entropy_models import pack_entropy_bottleneck
    return torch.stack([pack_entropy_bottleneck(dict(e.named_parameters())) for e in ebs], 0).contiguous()


def _fold_csc_into_cgp(convs, cs, G):
    """(packed, dims) of the cgp stack with the masked context conv folded into layer 0 (float64 on the host, once per
    weight version).  convs: the four cgp layers as lists over planes; cs: the csc MaskedConv2d per plane (mask applied);
    G subbands.  Layer 0 of subband g sees [plc_g (81), csc_g (81)] (LiftingBasedDWT_net.py:357-359)."""
    K = cs[0].kernel_size[0]
    bits_ = cs[0].tap_bits()
    live = [t for t in range(K * K) if (bits_ >> t) & 1]
    w0n, b0n = [], []
    for pl in range(len(cs)):
        W0 = convs[0][pl].weight.detach().double()[:, :, 0, 0]          # (G*c1, 2*cpl)
        b0 = convs[0][pl].bias.detach().double()
        Wc = cs[pl].weight.detach().double()                            # (G*cc, 1, K, K)
        bc = cs[pl].bias.detach().double()
        c1 = W0.shape[0] // G
        cc = Wc.shape[0] // G
        cpl = W0.shape[1] - cc
        rows_w, rows_b = [], []
        for g in range(G):
            W0g = W0[g * c1:(g + 1) * c1]                               # (c1, cpl + cc)
            Wcg = Wc[g * cc:(g + 1) * cc, 0].reshape(cc, K * K)[:, live]
            rows_w.append(torch.cat([W0g[:, :cpl], W0g[:, cpl:] @ Wcg], 1))
            rows_b.append(b0[g * c1:(g + 1) * c1] + W0g[:, cpl:] @ bc[g * cc:(g + 1) * cc])
        w0n.append(torch.cat(rows_w, 0).float()[:, :, None, None])
        b0n.append(torch.cat(rows_b, 0).float())
    ws = [torch.stack(w0n, 0).contiguous()] + [_stack(layer, lambda m: m.weight) for layer in convs[1:]]
    bs = [torch.stack(b0n, 0).contiguous()] + [_stack(layer, lambda m: m.bias) for layer in convs[1:]]
    packed, dims = ops.cgp_pack(ws, bs, G)
    packed16 = ops.cgp16_pack(ws, bs, G) if ops.cgp16_supported(ws, G) else None       # split-fp16 fragments (eval path)
    return packed, dims, packed16


def _fold_csc_train(cg, cs, xq, want_patches=True):
    """Differentiable version of _fold_csc_into_cgp for the training path.  Returns the folded layer-0 weight
    (P, G*c1, cpl + ntaps, 1, 1), bias (P, G*c1) and the gathered taps of xq as a (P,B,G*ntaps,h,w) tensor (None when
    want_patches is False: the kernels of CgpRateCtxFn gather them themselves)."""
    for m in cs:
        m.apply_mask_()
    K = cs[0].kernel_size[0]
    R = K // 2
    bits_ = cs[0].tap_bits()
    live = [t_ for t_ in range(K * K) if (bits_ >> t_) & 1]
    G = cg[0][0].groups
    W0 = _tstack([s_[0] for s_ in cg], lambda m: m.weight)[:, :, :, 0, 0]        # (P, G*c1, cpl + cc)
    b0 = _tstack([s_[0] for s_ in cg], lambda m: m.bias)                           # (P, G*c1)
    Wc = _tstack(cs, lambda m: m.weight)                                           # (P, G*cc, 1, K, K)
    bc = _tstack(cs, lambda m: m.bias)                                             # (P, G*cc)
    P = W0.shape[0]
    c1, cc = W0.shape[1] // G, Wc.shape[1] // G
    cpl = W0.shape[2] - cc
    W0g = W0.reshape(P, G, c1, cpl + cc)
    Wcg = Wc.reshape(P, G, cc, K * K)[:, :, :, live]                                # (P, G, cc, ntaps)
    Wf = torch.matmul(W0g[..., cpl:], Wcg)                                          # (P, G, c1, ntaps)
    w0f = torch.cat([W0g[..., :cpl], Wf], dim=3).reshape(P, G * c1, cpl + len(live), 1, 1)
    b0f = (b0.reshape(P, G, c1) + torch.matmul(W0g[..., cpl:], bc.reshape(P, G, cc, 1))[..., 0]).reshape(P, G * c1)
    if not want_patches:
        return w0f.contiguous(), b0f.contiguous(), None
    # taps of the quantised subband, zero outside the image: patches[:, :, g*ntaps + j] = xq[:, :, g] shifted by tap j
    Pn, B, Gx, h, w = xq.shape
    xp = F.pad(xq, (R, R, R, R))
    taps = [xp[:, :, :, (t_ // K):(t_ // K) + h, (t_ % K):(t_ % K) + w] for t_ in live]   # each (P,B,G,h,w)
    patches = torch.stack(taps, dim=3).reshape(Pn, B, Gx * len(live), h, w)
    return w0f.contiguous(), b0f.contiguous(), patches


def _stack5(seqs, t):
    for n in (0, 2, 4, 6, 8):
        t = _tconv([s_[n] for s_ in seqs], t, ops.ACT_NONE if n == 8 else ops.ACT_LRELU)
    return t


def _entropy_train_cond2(em, out_xe, out_xo, rnd):
    L = len(out_xo)
    xe_q = ag.QuantNoiseFn.apply(out_xe, rnd(out_xe))
    si_xe = ag.GaussRateFn.apply(out_xe, _stack5([l.csc_xe for l in em], xe_q), rnd(out_xe))
    q_list, si_list = [], []
    i = L - 1
    xo_q = ag.QuantNoiseFn.apply(out_xo[i], rnd(out_xo[i]))
    si_list.append(ag.GaussRateFn.apply(out_xo[i], _stack5([l.csc_list[i] for l in em], xo_q), rnd(out_xo[i])))
    q_list.append(xo_q)
    parent = xo_q
    for i in range(L - 2, -1, -1):
        xo_q = ag.QuantNoiseFn.apply(out_xo[i], rnd(out_xo[i]))
        seqs = [l.plc_list[i] for l in em]
        plc = _tconv([s_[2] for s_ in seqs], _tconv([s_[0] for s_ in seqs], parent, ops.ACT_LRELU, upsample2=True))
        cg = [l.cgp_out_xo_list[i] for l in em]
        cs = [l.csc_list[i] for l in em]
        # Same fold as the eval path (the masked csc conv is linear into cgp layer 0), written with differentiable tensor
        # ops on the PARAMETERS (tiny matmuls) and on the 12-tap patch gather (data movement), so autograd carries the
        # gradients of the fused kernels back to W0, the csc weights / bias and the quantised subband by itself.
        G = cg[0][0].groups
        no_cat = _CGP_TRAIN_CTX and xo_q.shape[-1] * xo_q.shape[-2] < (1 << 31)
        w0f, b0f, patches = _fold_csc_train(cg, cs, xo_q, want_patches=not no_cat)
        wb = [w0f, b0f] + [p for n in (2, 4, 6) for p in (_tstack([s_[n] for s_ in cg], lambda m: m.weight),
                                                          _tstack([s_[n] for s_ in cg], lambda m: m.bias))]
        if no_cat:      # the kernels read plc and gather the taps of xo_q themselves: no [plc_g | taps_g] tensor (1.75 GB at level 0)
            si_list.append(ag.CgpRateCtxFn.apply(plc.contiguous(), xo_q.contiguous(), out_xo[i], rnd(out_xo[i]), G,
                                                 cs[0].kernel_size[0], cs[0].tap_bits(), *wb))
        else:
            pl, pa = plc.chunk(G, dim=2), patches.chunk(G, dim=2)
            t = torch.cat([z_ for g_ in range(G) for z_ in (pl[g_], pa[g_])], dim=2)      # per subband [plc_g | taps_g]
            si_list.append(ag.CgpRateFn.apply(t.contiguous(), out_xo[i], rnd(out_xo[i]), G, *wb))
        q_list.append(xo_q)
        parent = xo_q
    q_list.reverse()
    si_list.reverse()
    return si_xe, si_list, xe_q, q_list


def _entropy_train_factorized(em, out_xe, out_xo, rnd):
    si_list, q_list = [], []
    for i in range(len(out_xo)):
        bits, q = ag.FactorizedRateFn.apply(out_xo[i], _eb_train([l.ent_out_xo_list[i] for l in em]), rnd(out_xo[i]))
        si_list.append(bits)
        q_list.append(q)
    si_xe, xe_q = ag.FactorizedRateFn.apply(out_xe, _eb_train([l.ent_out_xe for l in em]), rnd(out_xe))
    return si_xe, si_list, xe_q, q_list


def _entropy_train_ezwt(em, out_xe, out_xo, rnd):
    L = len(out_xo)
    si_xe, xe_q = ag.FactorizedRateFn.apply(out_xe, _eb_train([l.ent_out_xe for l in em]), rnd(out_xe))
    bits, q = ag.FactorizedRateFn.apply(out_xo[L - 1], _eb_train([l.ent_out_xo for l in em]), rnd(out_xo[L - 1]))
    si_list, q_list = [bits], [q]
    parent = q
    for i in range(L - 2, -1, -1):
        seqs = [l.plc_list[i] for l in em]
        t = _tconv([s_[0] for s_ in seqs], parent, ops.ACT_LRELU, upsample2=True)
        t = _tconv([s_[2] for s_ in seqs], t, ops.ACT_LRELU)
        ms = _tconv([s_[4] for s_ in seqs], t)
        noise = rnd(out_xo[i])
        si_list.append(ag.GaussRateFn.apply(out_xo[i], ms, noise))
        q = ag.QuantNoiseFn.apply(out_xo[i], noise)       # forward() returns x + the SAME noise as the rate (:832)
        q_list.append(q)
        parent = q
    q_list.reverse()
    si_list.reverse()
    return si_xe, si_list, xe_q, q_list


def _entropy_train_ztblock(em, out_xe, out_xo, rnd):
    L = len(out_xo)
    si_xe, xe_q = ag.FactorizedRateFn.apply(out_xe, _eb_train([l.ent_out_xe for l in em]), rnd(out_xe))
    bits, q = ag.FactorizedRateFn.apply(out_xo[L - 1], _eb_train([l.ent_out_xo for l in em]), rnd(out_xo[L - 1]))
    si_list, q_list = [bits], [q]
    con = q
    slots = ((0, 0), (0, 1), (1, 0), (1, 1))
    for i in range(L - 1):
        lev = L - i - 2
        x = out_xo[lev]
        P, B, _, H, W = x.shape
        sis, qs = [], []
        for j in range(3):
            xj = x[:, :, j:j + 1].contiguous()
            qj = ag.QuantNoiseFn.apply(xj, rnd(xj))
            ee, eo, oe = qj[..., 0::2, 0::2], qj[..., 0::2, 1::2], qj[..., 1::2, 0::2]
            dep1 = con[:, :, j:j + 1]
            deps = (dep1, torch.cat((dep1, ee), 2), torch.cat((dep1, ee, eo), 2), torch.cat((dep1, ee, eo, oe), 2))
            idx = j + i * 3
            rows = []
            for ch, kind in ((0, "sigma"), (1, "mu")):
                ph = []
                for k in range(4):
                    ph.append(_stack5([getattr(l, "dep_%d_list_%s" % (k + 1, kind))[idx] for l in em], deps[k].contiguous()))
                # interleave the 4 phases back to (H, W): data movement only
                top = torch.stack((ph[0], ph[1]), -1).flatten(-2)          # even rows: ee, eo interleaved along W
                bot = torch.stack((ph[2], ph[3]), -1).flatten(-2)
                rows.append(torch.stack((top, bot), -2).flatten(-3, -2))   # interleave rows
            params = torch.cat(rows, 2).contiguous()
            sis.append(ag.GaussRateFn.apply(xj, params, rnd(xj)))
            qs.append(qj)
        si_list.append(torch.cat(sis, 2))
        con = torch.cat(qs, 2)
        q_list.append(con)
    q_list.reverse()
    si_list.reverse()
    return si_xe, si_list, xe_q, q_list


def forward_planes_train(nets, x, noise_fn=None):
    """Differentiable encode -> entropy model (noise) -> decode for the plane nets (LiftingBasedDWT_net.py:154-170 in
    training mode; quirk 2 of SURVEY 8a: the context/decoder see noise sample #1, the rate an independent sample #2)."""
    aenc = [n.autoencoder for n in nets]
    em = [n.entropymodel for n in nets]

    def rnd(t):
        return torch.empty_like(t).uniform_(-0.5, 0.5) if noise_fn is None else noise_fn(t)

    lifting = isinstance(aenc[0], LiftingBasedNeuralWaveletv4)
    if lifting:
        L = aenc[0].waveletLevel
        taps, meta, Wt, nh, nl = _lift_params(aenc)
        outs = ag.LiftingFn.apply(x, taps, meta, nh, nl, *Wt)
        ll, yh = outs[0], list(outs[1:])
    else:
        L = aenc[0].dwtlevels
        outs = ag.Cdf97Fn.apply(x, L)
        ll = outs[0]
        yh = [t.reshape(t.shape[0], t.shape[1], -1, t.shape[4], t.shape[5]) for t in outs[1:]]
    out_xe = _ae_train([n.Yl_ae for n in aenc], ll, False)
    out_xo = [_ae_train([n.Yh_ae[i] for n in aenc], yh[i], False) for i in range(L)]
    fn = {DWTConditioned2EntropyLayerZTsepSubbands: _entropy_train_cond2, DWTFactorizedEntropyLayer: _entropy_train_factorized,
          onlyEZWT: _entropy_train_ezwt, DWTConditioned2EntropyLayerZTBlock: _entropy_train_ztblock}[type(em[0])]
    si_xe, si_list, xe_q, q_list = fn(em, out_xe, out_xo, rnd)
    Yl = _ae_train([n.Yl_ae for n in aenc], xe_q, True)
    Yh = [_ae_train([n.Yh_ae[i] for n in aenc], q_list[i], True) for i in range(L)]
    if lifting:
        xhat = ag.LiftingInvFn.apply(taps, meta, L, nh, nl, Yl, *Yh, *Wt)
    else:
        Yh6 = [t.reshape(t.shape[0], t.shape[1], t.shape[2] // 3, 3, t.shape[3], t.shape[4]) for t in Yh]
        xhat = ag.Cdf97InvFn.apply(Yl, *Yh6)
    return xhat, si_xe, si_list


_ENTROPY = {"factorized": DWTFactorizedEntropyLayer, "onlyEZWT": onlyEZWT,
            "conditioned2ZTsepSubbands": DWTConditioned2EntropyLayerZTsepSubbands,
            "DWTConditioned2EntropyLayerZTBlock": DWTConditioned2EntropyLayerZTBlock}
_TRANSFORM = {"CDF97": DWTPytorchWaveletsLayer, "LiftingBasedNeuralWaveletv4": LiftingBasedNeuralWaveletv4}


# ------------------------------------------------------------------------------------------------ nets
def forward_planes(nets, x, training):
    """encode -> entropymodel -> decode for a list of per-plane nets; x (P,B,C,H,W) plane-major.
    (LiftingBasedDWTNet.forward, LiftingBasedDWT_net.py:154-170)"""
    out_xe, out_xo = encode_planes([n.autoencoder for n in nets], x)
    em = [n.entropymodel for n in nets]
    si_xe, si_xo, xe_q, xo_q = type(em[0]).forward_planes(em, out_xe, out_xo, training)
    xhat = decode_planes([n.autoencoder for n in nets], xe_q, xo_q)
    return xhat, si_xe, si_xo


def rate_planes(nets, x, training=False):
    """The metric's path: lifting DWT encode + entropy-model forward (no decode) -> (si_xe, si_xo_list)."""
    out_xe, out_xo = encode_planes([n.autoencoder for n in nets], x)
    em = [n.entropymodel for n in nets]
    si_xe, si_xo, _, _ = type(em[0]).forward_planes(em, out_xe, out_xo, training)
    return si_xe, si_xo


def compress_planes(nets, x):
    """encode -> real entropy coding (compress + decompress from the strings) -> decode, for a list of per-plane nets;
    x (P,B,C,H,W) -> (xhat, strings_xe[p][b], [strings_xo[p][b] per level])."""
    em = [n.entropymodel for n in nets]
    if not hasattr(type(em[0]), "test_planes"):
        raise NotImplementedError("real entropy coding exists for conditioned2ZTsepSubbands (as in the reference) and, as an "
                                  "extension, onlyEZWT; the other entropy layers have no test() (LiftingBasedDWT_net.py:"
                                  "145-146 would fail there too)")
    out_xe, out_xo = encode_planes([n.autoencoder for n in nets], x)
    s_xe, s_xo, xe_q, xo_q = type(em[0]).test_planes(em, out_xe, out_xo)
    xhat = decode_planes([n.autoencoder for n in nets], xe_q, xo_q)
    return xhat, s_xe, s_xo


class LiftingBasedDWTNet(PackedOwnerMixin, nn.Module):
    def __init__(self, config):
        super().__init__()
        self._init_packed_owner()
        self.clrch = config.clrch
        if config.netType not in _TRANSFORM:
            raise ValueError("netType %r is not on the hot path (SURVEY.md 2: BasicWavelet/AttentionWavelet* are dead "
                             "or out of scope); supported: %s" % (config.netType, sorted(_TRANSFORM)))
        self.autoencoder = _TRANSFORM[config.netType](config)
        self.entropy_layer = config.entropy_layer
        if self.entropy_layer not in _ENTROPY:
            raise ValueError("entropy_layer %r not built yet; supported: %s" % (self.entropy_layer, sorted(_ENTROPY)))
        self.entropymodel = _ENTROPY[self.entropy_layer](config)

    def forward(self, x):
        xhat, si_xe, si_xo = forward_planes([self], x[None].contiguous(), self.training)
        return xhat[0], si_xe[0], [t[0] for t in si_xo]

    def compress(self, x):
        """LiftingBasedDWT_net.py:136-152: encode -> entropymodel.test (real coding, both directions) -> decode."""
        xhat, s_xe, s_xo = compress_planes([self], x[None].contiguous())
        one = lambda rows: rows[0][0] if len(rows[0]) == 1 else rows[0]
        return xhat[0], one(s_xe), [one(r) for r in s_xo]

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))


class LiftingBasedDWTNetWrapper(PackedOwnerMixin, nn.Module):
    def __init__(self, config):
        super().__init__()
        self._init_packed_owner()
        self.clrch = config.clrch
        if self.clrch == 3:
            self.model = LiftingBasedDWTNet(config)
        elif self.clrch == 1:
            self.model0 = LiftingBasedDWTNet(config)
            self.model1 = LiftingBasedDWTNet(config)
            self.model2 = LiftingBasedDWTNet(config)
        else:
            raise ValueError("clrch must be 1 or 3")

    def nets(self):
        return [self.model] if self.clrch == 3 else [self.model0, self.model1, self.model2]

    def forward_planes(self, x_pm):
        """Plane-major entry (P,B,C,H,W) -> plane-major (xhat, si_xe, [si_xo]); no layout copies."""
        return forward_planes(self.nets(), x_pm, self.training)

    def forward(self, x):
        """(B,3,H,W) -> (xhat (B,3,H,W), si_xe (B,3,h,w), list[3*L] of (B,3,h_i,w_i)), plane-major list order
        (LiftingBasedDWT_net.py:48-62)."""
        if self.clrch == 3:
            return self.model(x)
        x_pm = x.permute(1, 0, 2, 3).unsqueeze(2).contiguous()          # (3,B,1,H,W)
        xhat, si_xe, si_xo = self.forward_planes(x_pm)
        si_list = [t[p] for p in range(3) for t in si_xo]               # extend(): plane 0 levels, plane 1 ..., :58-61
        return (xhat[:, :, 0].permute(1, 0, 2, 3).contiguous(), si_xe[:, :, 0].permute(1, 0, 2, 3).contiguous(), si_list)

    def compress(self, x):
        """LiftingBasedDWT_net.py:76-99: -> (xhat, bpp of the low-pass streams, bpp of the subband streams); bytes * 8 over
        the pixels of the batch (the reference divides by H*W at its batch size of 1)."""
        if self.clrch == 3:
            return self.model.compress(x)
        B, _, H, W = x.shape
        x_pm = x.permute(1, 0, 2, 3).unsqueeze(2).contiguous()
        xhat, s_xe, s_xo = compress_planes(self.nets(), x_pm)
        self.last_strings = (s_xe, s_xo)
        len_xe = sum(byte_extractor(row) for row in s_xe)
        len_xo = sum(byte_extractor(row) for level in s_xo for row in level)
        return (xhat[:, :, 0].permute(1, 0, 2, 3).contiguous(), len_xe * 8 / (B * H * W), len_xo * 8 / (B * H * W))

    def aux_loss(self):
        return sum(n.aux_loss() for n in self.nets())
