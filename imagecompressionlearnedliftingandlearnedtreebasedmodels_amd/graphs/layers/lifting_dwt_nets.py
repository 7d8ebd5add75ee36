"""Transform layers of the hot path on the HIP kernels -- mirrors graphs/layers/lifting_dwt_nets.py of the reference:
``LiftingBasedNeuralWaveletv4`` (:646-827), ``SubbandAutoEncoder`` (:82-124), ``SubbandAutoEncoderBerk`` (:126-164),
``DWTPytorchWaveletsLayer`` (:212-277), ``get_cdf97_filters`` (:414-430), ``lifting_coeff`` (:431-432).

Modules are parameter containers with the reference's names / shapes / default initialisation (state_dict compatible,
SURVEY.md 8b).  The ``*_planes`` functions run a LIST of per-plane modules (the three colour-plane networks of
LiftingBasedDWTNetWrapper, LiftingBasedDWT_net.py:43-46) in single launches over plane-major (P,B,C,h,w) tensors.
"""
import torch
from torch import nn

from ... import ops
from ...packed_cache import PackedOwnerMixin, cached
from .P_block_v2 import P_block_v2
from .gdn import GDN
from .wavelet_forward_v2 import wavelet_forward_v2
from .wavelet_inverse_v2 import wavelet_inverse_v2

lifting_coeff = [-1.586134342059924, -0.052980118572961, 0.882911075530934, 0.443506852043971, 0.869864451624781,
                 1.149604398860241]  # bior4.4


def get_cdf97_filters(oned_or_twod="2D"):
    """The reference's CDF 9/7 filter table (lifting_dwt_nets.py:414-430); csrc/cdf97.hip holds the same taps."""
    lp = torch.tensor([0.0, 0.037828455507264, -0.023849465019557, -0.110624404418437, 0.377402855612831,
                       0.852698679008894, 0.377402855612831, -0.110624404418437, -0.023849465019557,
                       0.037828455507264], dtype=torch.double)
    hp = torch.tensor([0.0, -0.064538882628697, 0.040689417609164, 0.418092273221617, -0.788485616405583,
                       0.418092273221617, 0.040689417609164, -0.064538882628697, 0.0, 0.0], dtype=torch.double)
    slp = torch.tensor([0.0, -0.064538882628697, -0.040689417609164, 0.418092273221617, 0.788485616405583,
                        0.418092273221617, -0.040689417609164, -0.064538882628697, 0.0, 0.0], dtype=torch.double)
    shp = torch.tensor([0.0, -0.037828455507264, -0.023849465019557, 0.110624404418437, 0.377402855612831,
                        -0.852698679008894, 0.377402855612831, 0.110624404418437, -0.023849465019557,
                        -0.037828455507264], dtype=torch.double)
    if oned_or_twod == "1D":
        return lp, hp, slp, shp
    o = lambda a, b: a.view(-1, 1) * b.view(1, -1)
    return o(lp, lp), o(lp, hp), o(hp, lp), o(hp, hp), o(slp, slp), o(slp, shp), o(shp, slp), o(shp, shp)


# ------------------------------------------------------------------------------------------------ parameter cache
# stacked / packed device parameters live on the owning module (packed_cache.py): rebuilt when a source tensor changed


def _stack(mods, get):
    return torch.stack([get(m).detach() for m in mods], 0).contiguous()


# ------------------------------------------------------------------------------------------------ subband auto-encoders
class SubbandAutoEncoder(PackedOwnerMixin, nn.Module):
    """Per-coefficient scalar MLP (grouped 1x1 convs, tanh): lifting_dwt_nets.py:82-124."""

    def __init__(self, in_ch):
        super().__init__()
        self._init_packed_owner()
        iC, H = in_ch, 32
        self.in_ch, self.H = iC, H
        self.ae_down = nn.Sequential(
            nn.Conv2d(iC, iC * H, 1, groups=iC), nn.Tanh(), nn.Conv2d(iC * H, iC * H, 1, groups=iC), nn.Tanh(),
            nn.Conv2d(iC * H, iC * H, 1, groups=iC), nn.Tanh(), nn.Conv2d(iC * H, iC, 1, groups=iC))
        self.ae_up = nn.Sequential(
            nn.ConvTranspose2d(iC, iC * H, 1, groups=iC), nn.Tanh(), nn.ConvTranspose2d(iC * H, iC * H, 1, groups=iC),
            nn.Tanh(), nn.ConvTranspose2d(iC * H, iC * H, 1, groups=iC), nn.Tanh(),
            nn.ConvTranspose2d(iC * H, iC, 1, groups=iC))

    def encode(self, x):
        return ae_planes([self], x[None].contiguous(), False)[0]

    def decode(self, y_hat):
        return ae_planes([self], y_hat[None].contiguous(), True)[0]


class SubbandAutoEncoderBerk(PackedOwnerMixin, nn.Module):
    """3x3 convs + GDN (lifting_dwt_nets.py:126-164)."""

    def __init__(self, in_ch):
        super().__init__()
        self._init_packed_owner()
        iC, H, K, P = in_ch, 64, 3, 1
        self.in_ch = iC
        self.ae_down = nn.Sequential(
            nn.Conv2d(iC, iC * H // 2, K, padding=P), GDN(iC * H // 2), nn.Conv2d(iC * H // 2, iC * H, K, padding=P),
            GDN(iC * H), nn.Conv2d(iC * H, iC * H // 2, K, padding=P), GDN(iC * H // 2),
            nn.Conv2d(iC * H // 2, iC, K, padding=P))
        self.ae_up = nn.Sequential(
            nn.ConvTranspose2d(iC, iC * H // 2, K, padding=P), GDN(iC * H // 2, inverse=True),
            nn.ConvTranspose2d(iC * H // 2, iC * H, K, padding=P), GDN(iC * H, inverse=True),
            nn.ConvTranspose2d(iC * H, iC * H // 2, K, padding=P), GDN(iC * H // 2, inverse=True),
            nn.ConvTranspose2d(iC * H // 2, iC, K, padding=P))

    def encode(self, x):
        return ae_planes([self], x[None].contiguous(), False)[0]

    def decode(self, y_hat):
        return ae_planes([self], y_hat[None].contiguous(), True)[0]


def ae_planes(aes, x, decode):
    """Run the same-shaped subband auto-encoders ``aes`` (one per plane) on x (P,B,C,h,w)."""
    seq = [(a.ae_up if decode else a.ae_down) for a in aes]
    tag = ("ae", decode)
    own = aes[0]
    if isinstance(aes[0], SubbandAutoEncoder):
        convs = [[s[n] for s in seq] for n in (0, 2, 4, 6)]
        srcs = [p for layer in convs for m in layer for p in (m.weight, m.bias)]
        ws = cached(own, tag, srcs, lambda: [t for layer in convs for t in (
            _stack(layer, lambda m: m.weight).flatten(1), _stack(layer, lambda m: m.bias))])
        return ops.subband_mlp(x, *ws, transposed=decode, hidden=aes[0].H)
    t = x
    for n in (0, 2, 4, 6):
        layer = [s[n] for s in seq]
        w, b = cached(own, tag + (n,), [p for m in layer for p in (m.weight, m.bias)],
                          lambda: (_stack(layer, lambda m: m.weight), _stack(layer, lambda m: m.bias)))
        t = ops.conv2d(t, w, b, 3, transposed=decode)
        if n != 6:
            g = [s[n + 1] for s in seq]
            beta, gamma = cached(own, tag + (n, "g"), [p for m in g for p in (m.beta, m.gamma)],
                                     lambda: (_stack(g, lambda m: m.beta), _stack(g, lambda m: m.gamma)))
            t = ops.gdn(t, beta, gamma, inverse=decode, beta_min=g[0].beta_min)
    return t


# ------------------------------------------------------------------------------------------------ CDF 9/7 layer
class DWTPytorchWaveletsLayer(PackedOwnerMixin, nn.Module):
    """Fixed CDF 9/7 (bior4.4, periodization) + SubbandAutoEncoder per subband (lifting_dwt_nets.py:212-277)."""

    def __init__(self, config):
        super().__init__()
        self._init_packed_owner()
        self.dwtlevels = config.dwtlevels
        self.clrch = config.clrch
        self.Yl_ae = SubbandAutoEncoder(in_ch=1 * config.clrch)
        self.Yh_ae = nn.ModuleList([SubbandAutoEncoder(in_ch=3 * config.clrch) for _ in range(self.dwtlevels)])

    def encode(self, x):
        xe, xo = encode_planes([self], x[None].contiguous())
        return xe[0], [t[0] for t in xo]

    def decode(self, out_xe, out_xo_list):
        return decode_planes([self], out_xe[None].contiguous(), [t[None].contiguous() for t in out_xo_list])[0]


# ------------------------------------------------------------------------------------------------ learned lifting
class LiftingBasedNeuralWaveletv4(PackedOwnerMixin, nn.Module):
    """Learned lifting auto-encoder (lifting_dwt_nets.py:646-827)."""

    def __init__(self, config):
        super().__init__()
        self._init_packed_owner()
        self.waveletLevel = config.dwtlevels
        self.liftingLevel = config.num_lifting_perlayer
        self.blockprop = config.block_property
        self.clrch = config.clrch
        self.linearityflag = config.linearity_flag
        self.conv_filter_size = config.filtersize
        self.res_connection_weight = config.res_connection_weight
        self.config = config
        self.depth_scale = config.depth_scale * 8
        if self.clrch != 1 or self.liftingLevel != 2:
            raise ValueError("learned lifting needs clrch == 1 and num_lifting_perlayer == 2 "
                             "(lifting_dwt_nets.py:785-807 replaces the skip weights by (1,1,3,1) tensors; "
                             "wavelet_forward_v2.py:58-74 hard-codes two stages)")
        self.P_blocks = nn.ModuleList()
        self.U_blocks = nn.ModuleList()
        self.waveletForward = nn.ModuleList()
        self.waveletInverse = nn.ModuleList()
        self.Yh_ae = nn.ModuleList()
        self.preProcessingList = self.preProcessBlock(config.clrch, config.filtersize)
        ae = {"SubbandAutoEncoder": SubbandAutoEncoder, "SubbandAutoEncoderBerk": SubbandAutoEncoderBerk}[config.autoencoder]
        self.Yl_ae = ae(in_ch=1 * config.clrch)
        for _ in range(self.waveletLevel):
            self.Yh_ae.append(ae(in_ch=3 * config.clrch))
        self.nh = nn.Parameter(torch.zeros(1, 1, 1, 1))
        self.nl = nn.Parameter(torch.zeros(1, 1, 1, 1))
        nblocks = self.liftingLevel if self.blockprop == "same" else self.liftingLevel * 2 * self.waveletLevel
        for _ in range(nblocks):
            self.P_blocks.append(P_block_v2(self.linearityflag, self.clrch, self.conv_filter_size, self.depth_scale))
            self.U_blocks.append(P_block_v2(self.linearityflag, self.clrch, self.conv_filter_size, self.depth_scale))
        ll, wl = self.liftingLevel, self.waveletLevel
        for lev in range(wl):
            if self.blockprop == "same":
                fp, fu, ip, iu = self.P_blocks, self.U_blocks, self.P_blocks, self.U_blocks
            else:   # lifting_dwt_nets.py:711-722 (the inverse slices all start at wl*ll: SURVEY quirk 6)
                fp, fu = self.P_blocks[lev * ll:(lev + 1) * ll], self.U_blocks[lev * ll:(lev + 1) * ll]
                ip = self.P_blocks[wl * ll:wl * ll + (lev + 1) * ll]
                iu = self.U_blocks[wl * ll:wl * ll + (lev + 1) * ll]
            self.waveletForward.append(wavelet_forward_v2(fp, fu, self.res_connection_weight, ll, self.preProcessingList,
                                                          config, self.nh, self.nl, owner=self, level=lev))
            self.waveletInverse.append(wavelet_inverse_v2(ip, iu, self.res_connection_weight, ll, self.preProcessingList,
                                                          config, self.nh, self.nl, owner=self, level=lev))

    def preProcessBlock(self, csize, conv_filter_size):
        """Four learnable 3x1 skip filters initialised to the CDF 9/7 lifting constants (lifting_dwt_nets.py:784-827)."""
        a, b, g, d = lifting_coeff[:4]
        convs = nn.ModuleList()
        for taps in ([0.0, a, a], [b, b, 0.0], [0.0, g, g], [d, d, 0.0]):
            c = nn.Conv2d(csize, csize, kernel_size=(3, 1), stride=1, padding=(1, 0), bias=False)
            c.weight = nn.Parameter(torch.tensor(taps).view(1, 1, 3, 1))
            convs.append(c)
        return convs

    def encode(self, input):
        xe, xo = encode_planes([self], input[None].contiguous())
        return xe[0], [t[0] for t in xo]

    def decode(self, out_xe, out_xo_list):
        return decode_planes([self], out_xe[None].contiguous(), [t[None].contiguous() for t in out_xo_list])[0]


def _lifting_params(nets):
    """-> taps (4,P,3), packed (P,nblocks,2,total), nh/nl (P,) or None."""
    n0 = nets[0]
    nb = len(n0.P_blocks)
    srcs = [p for n in nets for p in n.parameters()]

    def build():
        taps = torch.stack([_stack(nets, lambda n, j=j: n.preProcessingList[j].weight.reshape(3)) for j in range(4)], 0)
        blocks = []
        for b in range(nb):
            pu = []
            for kind in ("P_blocks", "U_blocks"):
                args = []
                for cn in ("conv1", "conv2", "conv3", "conv4"):
                    args.append(_stack(nets, lambda n, k=kind, c=cn: getattr(getattr(n, k)[b], c).weight))
                    args.append(_stack(nets, lambda n, k=kind, c=cn: getattr(getattr(n, k)[b], c).bias))
                pu.append(ops.pack_pblock(*args))
            blocks.append(torch.stack(pu, 1))
        packed = torch.stack(blocks, 1).contiguous()
        nh = nl = None
        if n0.config.scale == 1:
            nh = _stack(nets, lambda n: lifting_coeff[4] + n.nh.reshape(()) * 0.1)
            nl = _stack(nets, lambda n: lifting_coeff[5] + n.nl.reshape(()) * 0.1)
        return taps.contiguous(), packed, nh, nl
    return cached(n0, ("lift", tuple(id(n) for n in nets[1:])), srcs, build)


def lifting_forward_planes(nets, x, levels=None, first_level=0):
    """x (P,B,1,H,W) -> (ll, [yh]) through lldwt_lifting_forward (lifting_dwt_nets.py:728-732)."""
    n0 = nets[0]
    taps, packed, nh, nl = _lifting_params(nets)
    different = n0.blockprop != "same"
    levels = n0.waveletLevel if levels is None else levels
    return ops.lifting_forward(x, taps, packed, levels, n0.depth_scale, n0.conv_filter_size, n0.res_connection_weight,
                               n0.linearityflag != 1, different, nh, nl,
                               block_offset=2 * first_level if different else 0)


def lifting_inverse_planes(nets, ll, yh, first_level=0):
    n0 = nets[0]
    taps, packed, nh, nl = _lifting_params(nets)
    # 'different': every inverse level uses the pair starting at waveletLevel*liftingLevel (lifting_dwt_nets.py:718-722)
    off = 2 * n0.waveletLevel if n0.blockprop != "same" else 0
    return ops.lifting_inverse(ll, yh, taps, packed, n0.depth_scale, n0.conv_filter_size, n0.res_connection_weight,
                               n0.linearityflag != 1, nh, nl, block_offset=off)


def encode_planes(nets, x):
    """autoencoder.encode for a list of per-plane transform modules; x (P,B,C,H,W) -> (out_xe, out_xo_list)."""
    n0 = nets[0]
    if isinstance(n0, DWTPytorchWaveletsLayer):
        ll, yh6 = ops.cdf97_forward(x, n0.dwtlevels)                       # yh (P,B,C,3,h,w)
        yh = [t.reshape(t.shape[0], t.shape[1], -1, t.shape[4], t.shape[5]) for t in yh6]   # view(B,C*3,h,w), :255-256
    else:
        ll, yh = lifting_forward_planes(nets, x)
    out_xe = ae_planes([n.Yl_ae for n in nets], ll, False)
    out_xo = [ae_planes([n.Yh_ae[i] for n in nets], yh[i], False) for i in range(len(yh))]
    return out_xe, out_xo


def decode_planes(nets, out_xe, out_xo_list):
    """autoencoder.decode for a list of per-plane transform modules -> xhat (P,B,C,H,W)."""
    n0 = nets[0]
    Yl = ae_planes([n.Yl_ae for n in nets], out_xe, True)
    Yh = [ae_planes([n.Yh_ae[i] for n in nets], out_xo_list[i], True) for i in range(len(out_xo_list))]
    if isinstance(n0, DWTPytorchWaveletsLayer):
        Yh6 = [t.reshape(t.shape[0], t.shape[1], t.shape[2] // 3, 3, t.shape[3], t.shape[4]) for t in Yh]
        return ops.cdf97_inverse(Yl, Yh6)
    return lifting_inverse_planes(nets, Yl, Yh)
