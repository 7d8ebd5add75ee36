"""Deterministic by-name weight filler + state-dict templates -- test infrastructure only.

The templates reproduce the parameter/buffer names and shapes of the reference modules
(graphs/layers/lifting_dwt_nets.py:646-722,784-827, graphs/models/LiftingBasedDWT_net.py:238-318,
:186-211, :786-801) for ONE per-plane net (``autoencoder.*`` + ``entropymodel.*``), unique tensors only (the
reference's state_dict additionally aliases the shared blocks under waveletForward/waveletInverse).
``fill_by_name`` overwrites every learnable tensor with values drawn from a generator seeded by crc32(key), so the
same numbers can be produced for the reference modules (tests/golden/make_golden.py), the oracle and the HIP path.
"""
import zlib

import torch

from .entropy import conv_mask, eb_init_state
from .lifting import LIFTING_COEFF
from .subband_ae import nonneg_init


def _conv(sd, name, cout, cin_g, kh, kw, bias=True):
    sd[name + ".weight"] = torch.zeros(cout, cin_g, kh, kw)
    if bias:
        sd[name + ".bias"] = torch.zeros(cout)


def _masked(sd, name, mtype, cin, cout, k, groups):
    _conv(sd, name, cout, cin // groups, k, k)
    sd[name + ".mask"] = conv_mask((cout, cin // groups, k, k), mtype)


def autoencoder_template(cfg):
    sd = {}
    c = cfg["clrch"]
    k = cfg["filtersize"]
    d = cfg["depth_scale"] * 8
    L = cfg["dwtlevels"]
    lifting = cfg["netType"] != "CDF97"
    if lifting:
        taps = [[0.0, LIFTING_COEFF[0], LIFTING_COEFF[0]], [LIFTING_COEFF[1], LIFTING_COEFF[1], 0.0],
                [0.0, LIFTING_COEFF[2], LIFTING_COEFF[2]], [LIFTING_COEFF[3], LIFTING_COEFF[3], 0.0]]
        for j, t in enumerate(taps):
            sd["preProcessingList.%d.weight" % j] = torch.tensor(t).view(1, 1, 3, 1)
        sd["nh"] = torch.zeros(1, 1, 1, 1)
        sd["nl"] = torch.zeros(1, 1, 1, 1)
        nb = cfg["num_lifting_perlayer"] * (1 if cfg["block_property"] == "same" else 2 * L)
        for b in range(nb):
            for kind in ("P_blocks", "U_blocks"):
                p = "%s.%d." % (kind, b)
                _conv(sd, p + "conv1", d * c, 1 * c, k, k)
                _conv(sd, p + "conv2", d * c, d * c, k, k)
                _conv(sd, p + "conv3", d * c, d * c, k, k)
                _conv(sd, p + "conv4", 1 * c, d * c, k, k)
    kind = cfg["autoencoder"] if lifting else "SubbandAutoEncoder"
    for name, ic in [("Yl_ae", c)] + [("Yh_ae.%d" % i, 3 * c) for i in range(L)]:
        if kind == "SubbandAutoEncoder":
            H = 32
            chans = [(ic, ic * H), (ic * H, ic * H), (ic * H, ic * H), (ic * H, ic)]
            for n, (a, b) in zip((0, 2, 4, 6), chans):
                _conv(sd, "%s.ae_down.%d" % (name, n), b, a // ic, 1, 1)
                # ConvTranspose2d weight is (in, out/groups, k, k)
                sd["%s.ae_up.%d.weight" % (name, n)] = torch.zeros(a, b // ic, 1, 1)
                sd["%s.ae_up.%d.bias" % (name, n)] = torch.zeros(b)
        else:
            H = 64
            chans = [(ic, ic * H // 2), (ic * H // 2, ic * H), (ic * H, ic * H // 2), (ic * H // 2, ic)]
            for n, (a, b) in zip((0, 2, 4, 6), chans):
                _conv(sd, "%s.ae_down.%d" % (name, n), b, a, 3, 3)
                sd["%s.ae_up.%d.weight" % (name, n)] = torch.zeros(a, b, 3, 3)
                sd["%s.ae_up.%d.bias" % (name, n)] = torch.zeros(b)
                if n != 6:
                    for ud in ("ae_down", "ae_up"):
                        sd["%s.%s.%d.beta" % (name, ud, n + 1)] = nonneg_init(torch.ones(b))
                        sd["%s.%s.%d.gamma" % (name, ud, n + 1)] = nonneg_init(0.1 * torch.eye(b))
    return sd


def entropy_template(cfg, gen=None):
    sd = {}
    L = cfg["dwtlevels"]
    c = cfg["clrch"]
    so, se = 3 * c, c
    layer = cfg["entropy_layer"]

    def eb(prefix, ch):
        for k, v in eb_init_state(ch, gen).items():
            sd[prefix + k] = v

    if layer == "factorized":
        for i in range(L):
            eb("ent_out_xo_list.%d." % i, so)
            sd["scl_out_xo_list.%d" % i] = torch.full((1, so, 1, 1), i + 1.0)
            sd["scb_out_xo_list.%d" % i] = torch.full((1, so, 1, 1), 1.0)
        eb("ent_out_xe.", se)
        sd["scl_out_xe"] = torch.full((1, se, 1, 1), 5.0)
        sd["scb_out_xe"] = torch.full((1, se, 1, 1), 0.2)
    elif layer == "onlyEZWT":
        for i in range(L - 1):
            _conv(sd, "plc_list.%d.0" % i, so * 81, so, 3, 3)
            _conv(sd, "plc_list.%d.2" % i, so * 81, so * 81, 3, 3)
            _conv(sd, "plc_list.%d.4" % i, 6, so * 81, 1, 1)
        eb("ent_out_xe.", 1)
        eb("ent_out_xo.", 3)
    elif layer == "conditioned2ZTsepSubbands":
        for i in range(L - 1):
            o1 = so * 81
            _conv(sd, "plc_list.%d.0" % i, o1, so, 3, 3)
            _conv(sd, "plc_list.%d.2" % i, o1, o1, 3, 3)
            _masked(sd, "csc_list.%d" % i, "A", so, so * 81, 5, so)
            inn = 2 * o1
            for n, (a, b) in zip((0, 2, 4, 6), [(inn, inn), (inn, inn // 3), (inn // 3, inn // 9), (inn // 9, so * 2)]):
                _conv(sd, "cgp_out_xo_list.%d.%d" % (i, n), b, a // so, 1, 1)

        def stack(prefix, g):
            o = g * 81
            for n, (t, a, b) in zip((0, 2, 4, 6, 8), [("A", g, o), ("B", o, o), ("B", o, o // 3), ("B", o // 3, o // 9),
                                                       ("B", o // 9, g * 2)]):
                _masked(sd, "%s.%d" % (prefix, n), t, a, b, 3, g)
        stack("csc_list.%d" % (L - 1), so)
        stack("csc_xe", se)
    elif layer == "DWTConditioned2EntropyLayerZTBlock":
        hid = 32
        for i in range(L - 1):
            for j in range(3):
                n = j + i * 3
                sd["scl_out_xo_list.%d" % n] = torch.full((1, so, 1, 1), i * 1.0 + 1.0)
                sd["scb_out_xo_list.%d" % n] = torch.full((1, so, 1, 1), 1.0)
                for k in range(1, 5):
                    for kind in ("mu", "sigma"):
                        pre = "dep_%d_list_%s.%d" % (k, kind, n)
                        for idx, (a, b, ks) in zip((0, 2, 4, 6, 8), [(k, hid, 3), (hid, hid, 3), (hid, hid, 1),
                                                                     (hid, hid, 1), (hid, 1, 1)]):
                            _conv(sd, "%s.%d" % (pre, idx), b, a, ks, ks)
        eb("ent_out_xe.", 1)
        eb("ent_out_xo.", 3)
    else:
        raise ValueError(layer)
    return sd


def net_template(cfg, gen=None):
    sd = {"autoencoder." + k: v for k, v in autoencoder_template(cfg).items()}
    sd.update({"entropymodel." + k: v for k, v in entropy_template(cfg, gen).items()})
    return sd


def wrapper_template(cfg, gen=None):
    sd = {}
    for c in range(3):
        for k, v in net_template(cfg, gen).items():
            sd["model%d.%s" % (c, k)] = v.clone()
    return sd


def fill_value(name, tensor, scale=1.0):
    """Deterministic value for one tensor, keyed by its (full) state_dict name."""
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    leaf = name.split(".")[-1]
    shape = tuple(tensor.shape)

    def u(*shp):                      # U(-0.5, 0.5)
        return torch.rand(shp if shp else shape, generator=g) - 0.5

    if leaf in ("mask", "target", "bound", "pedestal") or "scl_out" in name or "scb_out" in name:
        return tensor.clone()
    if leaf == "quantiles":           # perturb the medians only
        return tensor.clone() + 0.3 * u(shape[0], 1, 1) * torch.tensor([0.0, 1.0, 0.0])
    if leaf.startswith("_matrix"):
        return tensor.clone() + 0.1 * u()
    if leaf.startswith("_factor"):
        return 0.5 * u()
    if leaf.startswith("_bias"):
        return u()
    if leaf == "beta":
        return tensor.clone() * (1.0 + 0.2 * u())
    if leaf == "gamma":
        return tensor.clone() * (1.0 + 0.2 * u()) + 0.01 * (u() + 0.5)
    if "preProcessingList" in name:
        return tensor.clone() * (1.0 + 0.05 * u())
    if leaf in ("nh", "nl"):
        return 0.2 * u()
    if leaf == "weight":
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        if "ae_up" in name:           # ConvTranspose2d weight is (in, out/groups, kh, kw)
            fan_in = 32 if shape[2] == 1 else shape[0] * shape[2] * shape[3]
        return 2.0 * u() * scale * (3.0 / fan_in) ** 0.5
    if leaf == "bias":
        return 0.2 * u()
    raise KeyError("no fill rule for " + name)


def fill_by_name(sd, scale=1.0):
    """In-place deterministic fill of a (possibly prefixed) state dict; returns it."""
    for k in sorted(sd):
        v = fill_value(k, sd[k], scale)
        sd[k] = v.to(sd[k].dtype).reshape(sd[k].shape).contiguous()
    return sd
