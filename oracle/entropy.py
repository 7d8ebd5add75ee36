"""Oracle: entropy models (rate estimation) -- CPU restatement, test infrastructure only.

Wiring follows (paths relative to /root/reference):
  graphs/layers/masked_conv2d.py:5-21                      MaskedConv2d (mask A/B, weight *= mask)
  graphs/models/LiftingBasedDWT_net.py:182-231             DWTFactorizedEntropyLayer
  graphs/models/LiftingBasedDWT_net.py:233-372             DWTConditioned2EntropyLayerZTsepSubbands
  graphs/models/LiftingBasedDWT_net.py:759-840             onlyEZWT

Leaf ops restate compressai==1.2.1 (requirements.txt:2; package absent from the image and not vendored, the
reference holds no test vectors for it -> **parity unpinned** for these; call sites
LiftingBasedDWT_net.py:204,209,225,229,291,307,318,330,334,341,345,352,364,800-801,815,818,832):
  GaussianConditional.quantize / forward / _likelihood / _standardized_cumulative
  EntropyBottleneck.forward / _logits_cumulative / _likelihood / _get_medians
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .subband_ae import lower_bound

SCALE_BOUND = 0.11          # GaussianConditional(scale_bound=0.11), LiftingBasedDWT_net.py:291,307,318
LIKELIHOOD_BOUND = 1e-9     # compressai EntropyModel default likelihood_bound
EB_FILTERS = (1, 3, 3, 3, 3, 1)
EB_INIT_SCALE = 10.0
EB_TAIL_MASS = 1e-9


# ----------------------------------------------------------------------------- leaf ops
def quantize(x, mode, means=None, noise=None):
    """compressai EntropyModel.quantize: 'noise' -> x + U(-.5,.5); 'dequantize' -> round(x-mu)+mu."""
    if mode == "noise":
        if noise is None:
            noise = torch.empty_like(x).uniform_(-0.5, 0.5)
        return x + noise
    out = x.clone()
    if means is not None:
        out = out - means
    out = torch.round(out)
    if means is not None:
        out = out + means
    return out


def std_cumulative(z):
    """0.5 * erfc(-z / sqrt(2))."""
    return 0.5 * torch.erfc(-(2 ** -0.5) * z)


def gaussian_likelihood(v, scales, means=None):
    """GaussianConditional._likelihood + likelihood_lower_bound."""
    values = v - means if means is not None else v
    s = lower_bound(scales, SCALE_BOUND)
    values = torch.abs(values)
    upper = std_cumulative((0.5 - values) / s)
    lower = std_cumulative((-0.5 - values) / s)
    return lower_bound(upper - lower, LIKELIHOOD_BOUND)


def gaussian_conditional_forward(x, scales, means, training, noise=None):
    """GaussianConditional.forward(x, scales, means=..., training=...) -> (outputs, likelihood)."""
    out = quantize(x, "noise" if training else "dequantize", means, noise)
    return out, gaussian_likelihood(out, scales, means)


def eb_logits_cumulative(v, sd, prefix):
    """EntropyBottleneck._logits_cumulative; v: (C,1,N)."""
    logits = v
    for i in range(len(EB_FILTERS) - 1):
        logits = torch.matmul(F.softplus(sd[prefix + "_matrix%d" % i]), logits)
        logits = logits + sd[prefix + "_bias%d" % i]
        if i < len(EB_FILTERS) - 2:
            logits = logits + torch.tanh(sd[prefix + "_factor%d" % i]) * torch.tanh(logits)
    return logits


def eb_likelihood(v, sd, prefix):
    lower = eb_logits_cumulative(v - 0.5, sd, prefix)
    upper = eb_logits_cumulative(v + 0.5, sd, prefix)
    sign = -torch.sign(lower + upper).detach()
    lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
    return lower_bound(lik, LIKELIHOOD_BOUND)


def entropy_bottleneck_forward(x, sd, prefix, training, noise=None):
    """EntropyBottleneck.forward -> (outputs, likelihood); x: (B,C,H,W)."""
    B, C = x.shape[:2]
    xp = x.transpose(0, 1).contiguous()
    shape = xp.shape
    values = xp.reshape(C, 1, -1)
    medians = sd[prefix + "quantiles"][:, :, 1:2].detach()
    if noise is not None:
        noise = noise.transpose(0, 1).reshape(C, 1, -1)
    outputs = quantize(values, "noise" if training else "dequantize", medians, noise)
    lik = eb_likelihood(outputs, sd, prefix)
    outputs = outputs.reshape(shape).transpose(0, 1).contiguous()
    lik = lik.reshape(shape).transpose(0, 1).contiguous()
    return outputs, lik


def eb_init_state(channels, gen=None):
    """EntropyBottleneck.__init__ parameter initialisation (compressai 1.2.1): returns dict of tensors."""
    sd = {}
    nf = len(EB_FILTERS) - 1
    scale = EB_INIT_SCALE ** (1.0 / nf)
    for i in range(nf):
        init = float(np.log(np.expm1(1.0 / scale / EB_FILTERS[i + 1])))
        sd["_matrix%d" % i] = torch.full((channels, EB_FILTERS[i + 1], EB_FILTERS[i]), init)
        sd["_bias%d" % i] = torch.empty(channels, EB_FILTERS[i + 1], 1).uniform_(-0.5, 0.5, generator=gen)
        if i < nf - 1:
            sd["_factor%d" % i] = torch.zeros(channels, EB_FILTERS[i + 1], 1)
    sd["quantiles"] = torch.tensor([-EB_INIT_SCALE, 0.0, EB_INIT_SCALE]).repeat(channels, 1, 1)
    t = math.log(2.0 / EB_TAIL_MASS - 1.0)
    sd["target"] = torch.tensor([-t, 0.0, t])
    return sd


# ----------------------------------------------------------------------------- masked conv
def conv_mask(weight_shape, mask_type):
    """graphs/layers/masked_conv2d.py:9-17."""
    _, _, kH, kW = weight_shape
    mask = torch.ones(weight_shape)
    b = 1 if mask_type == "B" else 0
    if kW > 1:
        mask[:, :, kH // 2, kW // 2 + b:] = 0
    elif kW == 1 and mask_type == "A":
        mask[:, :, kH // 2, kW // 2 + b:] = 0
    if kH > 1:
        mask[:, :, kH // 2 + 1:] = 0
    return mask


def masked_conv(x, sd, prefix, groups):
    """MaskedConv2d.forward (masked_conv2d.py:19-21): weight*mask then conv (mask is a registered buffer)."""
    w = sd[prefix + "weight"] * sd[prefix + "mask"]
    return F.conv2d(x, w, sd[prefix + "bias"], padding=w.shape[-1] // 2, groups=groups)


def _csc_stack(x, sd, prefix, groups):
    """5-layer masked 3x3 stack (LiftingBasedDWT_net.py:299-305, :311-317), LeakyReLU(0.01) between."""
    t = x
    for n in (0, 2, 4, 6, 8):
        t = masked_conv(t, sd, prefix + "%d." % n, groups)
        if n != 8:
            t = F.leaky_relu(t, 0.01)
    return t


def upsample2(x):
    """repeat_interleave(2,dim=2).repeat_interleave(2,dim=3) (LiftingBasedDWT_net.py:348,367,822,835)."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def _neg_log2(p):
    return -torch.log2(p)


# ----------------------------------------------------------------------------- entropy layers
def factorized_forward(out_xe, out_xo_list, sd, cfg, training=False, noises=None, dbg=None):
    """DWTFactorizedEntropyLayer.forward (LiftingBasedDWT_net.py:215-231)."""
    nlev = cfg["dwtlevels"]
    si_xo, q_xo = [], []
    for i in range(nlev):
        q, p = entropy_bottleneck_forward(out_xo_list[i], sd, "ent_out_xo_list.%d." % i, training,
                                          None if noises is None else noises["xo"][i])
        si_xo.append(_neg_log2(p))
        q_xo.append(q)
    q_xe, p = entropy_bottleneck_forward(out_xe, sd, "ent_out_xe.", training, None if noises is None else noises["xe"])
    return _neg_log2(p), si_xo, q_xe, q_xo


def conditioned2_forward(out_xe, out_xo_list, sd, cfg, training=False, noises=None, dbg=None):
    """DWTConditioned2EntropyLayerZTsepSubbands.forward (LiftingBasedDWT_net.py:322-372).

    ``dbg`` (tests only): a dict that receives the rate-domain residuals x - mu per tensor ('xe', level i), so a test can
    tell a rounding flip of round(x - mu) (a residual within float noise of a half-integer) from a real mismatch.

    ``noises`` (training only, for reproducible tests): dict with 'xe': (n1, n2), 'xo': [(n1, n2)] -- n1 is the
    context/decoder sample (:330,341,352), n2 the independent sample drawn inside forward (:334,345,364)."""
    nlev = cfg["dwtlevels"]
    mode = "noise" if training else "dequantize"

    def nz(key, i=None, k=0):
        if noises is None:
            return None
        return noises[key][k] if i is None else noises[key][i][k]

    # xe (:330-335).  Quirk: the decoder gets quantize(x) WITHOUT means; the rate is evaluated inside forward.
    xe_q = quantize(out_xe, mode, None, nz("xe", None, 0))
    ms = _csc_stack(xe_q, sd, "csc_xe.", groups=out_xe.shape[1])
    sigma, mu = ms[:, 0::2], ms[:, 1::2]
    if dbg is not None:
        dbg["xe"] = out_xe - mu
    _, p = gaussian_conditional_forward(out_xe, sigma, mu, training, nz("xe", None, 1))
    si_xe = _neg_log2(p)

    q_list, si_list = [], []
    i = nlev - 1
    xo_q = quantize(out_xo_list[i], mode, None, nz("xo", i, 0))
    ms = _csc_stack(xo_q, sd, "csc_list.%d." % i, groups=out_xo_list[i].shape[1])
    sigma, mu = ms[:, 0::2], ms[:, 1::2]
    if dbg is not None:
        dbg[i] = out_xo_list[i] - mu
    _, p = gaussian_conditional_forward(out_xo_list[i], sigma, mu, training, nz("xo", i, 1))
    si_list.append(_neg_log2(p))
    q_list.append(xo_q)
    con = upsample2(xo_q)
    for i in range(nlev - 2, -1, -1):
        g = out_xo_list[i].shape[1]
        xo_q = quantize(out_xo_list[i], mode, None, nz("xo", i, 0))
        csc = masked_conv(xo_q, sd, "csc_list.%d." % i, groups=g)                       # :353
        plc = F.conv2d(con, sd["plc_list.%d.0.weight" % i], sd["plc_list.%d.0.bias" % i], padding=1)
        plc = F.leaky_relu(plc, 0.01)
        plc = F.conv2d(plc, sd["plc_list.%d.2.weight" % i], sd["plc_list.%d.2.bias" % i], padding=1)   # :355
        p0, p1, p2 = plc.chunk(3, dim=1)
        c0, c1, c2 = csc.chunk(3, dim=1)
        t = torch.cat((p0, c0, p1, c1, p2, c2), dim=1)                                    # :357-359
        for n in (0, 2, 4, 6):                                                            # :360 (cgp, groups=inn_ch1)
            t = F.conv2d(t, sd["cgp_out_xo_list.%d.%d.weight" % (i, n)], sd["cgp_out_xo_list.%d.%d.bias" % (i, n)],
                         groups=g)
            if n != 6:
                t = F.leaky_relu(t, 0.01)
        sigma, mu = t[:, 0::2], t[:, 1::2]
        if dbg is not None:
            dbg[i] = out_xo_list[i] - mu
        _, p = gaussian_conditional_forward(out_xo_list[i], sigma, mu, training, nz("xo", i, 1))
        si_list.append(_neg_log2(p))
        q_list.append(xo_q)
        con = upsample2(xo_q)
    q_list.reverse()
    si_list.reverse()
    return si_xe, si_list, xe_q, q_list


def only_ezwt_forward(out_xe, out_xo_list, sd, cfg, training=False, noises=None, dbg=None):
    """onlyEZWT.forward (LiftingBasedDWT_net.py:804-840)."""
    nlev = cfg["dwtlevels"]
    q_list, si_list = [], []
    xe_q, p = entropy_bottleneck_forward(out_xe, sd, "ent_out_xe.", training, None if noises is None else noises["xe"])
    si_xe = _neg_log2(p)
    i = nlev - 1
    xo_q, p = entropy_bottleneck_forward(out_xo_list[i], sd, "ent_out_xo.", training,
                                         None if noises is None else noises["xo"][i])
    si_list.append(_neg_log2(p))
    q_list.append(xo_q)
    con = upsample2(xo_q)
    for i in range(nlev - 2, -1, -1):
        t = F.conv2d(con, sd["plc_list.%d.0.weight" % i], sd["plc_list.%d.0.bias" % i], padding=1)
        t = F.leaky_relu(t, 0.01)
        t = F.conv2d(t, sd["plc_list.%d.2.weight" % i], sd["plc_list.%d.2.bias" % i], padding=1)
        t = F.leaky_relu(t, 0.01)
        t = F.conv2d(t, sd["plc_list.%d.4.weight" % i], sd["plc_list.%d.4.bias" % i])
        sigma, mu = t[:, 0::2], t[:, 1::2]
        if dbg is not None:
            dbg[i] = out_xo_list[i] - mu
        xo_q, p = gaussian_conditional_forward(out_xo_list[i], sigma, mu, training,
                                               None if noises is None else noises["xo"][i])
        si_list.append(_neg_log2(p))
        q_list.append(xo_q)
        con = upsample2(xo_q)
    q_list.reverse()
    si_list.reverse()
    return si_xe, si_list, xe_q, q_list


def _dep_net(x, sd, prefix):
    """5-layer CNN of DWTConditioned2EntropyLayerZTBlock (LiftingBasedDWT_net.py:618-624): 3x3, 3x3, 1x1, 1x1, 1x1."""
    t = x
    for n in (0, 2, 4, 6, 8):
        w = sd[prefix + "%d.weight" % n]
        t = F.conv2d(t, w, sd[prefix + "%d.bias" % n], padding=w.shape[-1] // 2)
        if n != 8:
            t = F.leaky_relu(t, 0.01)
    return t


def ztblock_forward(out_xe, out_xo_list, sd, cfg, training=False, noises=None, dbg=None):
    """DWTConditioned2EntropyLayerZTBlock.forward (LiftingBasedDWT_net.py:691-757): the four polyphase phases of every
    subband are predicted in sequence from the (not upsampled) parent and the phases already coded.
    noises (training): {'xe': n, 'xo_top': n, 'xo': [[(n1,n2) per subband j] per level]}."""
    L = cfg["dwtlevels"]
    mode = "noise" if training else "dequantize"
    xe_q, p = entropy_bottleneck_forward(out_xe, sd, "ent_out_xe.", training, None if noises is None else noises["xe"])
    si_xe = _neg_log2(p)
    xo_q, p = entropy_bottleneck_forward(out_xo_list[L - 1], sd, "ent_out_xo.", training,
                                         None if noises is None else noises["xo_top"])
    si_list, q_list = [_neg_log2(p)], [xo_q]
    con = xo_q
    for i in range(L - 1):
        lev = L - i - 2
        x = out_xo_list[lev]
        B, C, H, W = x.shape
        sis, qs = [], []
        for j in range(3):
            xj = x[:, j:j + 1]
            n1 = None if noises is None else noises["xo"][lev][j][0]
            n2 = None if noises is None else noises["xo"][lev][j][1]
            q = quantize(xj, mode, None, n1)                                             # :716-718
            ee, eo, oe = q[:, :, 0::2, 0::2], q[:, :, 0::2, 1::2], q[:, :, 1::2, 0::2]   # :719-721
            dep1 = con[:, j:j + 1]
            mu = torch.empty(B, 1, H, W)
            sg = torch.empty(B, 1, H, W)
            idx = j + i * 3
            deps = [dep1, torch.cat((dep1, ee), 1), torch.cat((dep1, ee, eo), 1), torch.cat((dep1, ee, eo, oe), 1)]
            slots = [(slice(0, None, 2), slice(0, None, 2)), (slice(0, None, 2), slice(1, None, 2)),
                     (slice(1, None, 2), slice(0, None, 2)), (slice(1, None, 2), slice(1, None, 2))]
            for k in range(4):                                                            # :723-740
                mu[:, :, slots[k][0], slots[k][1]] = _dep_net(deps[k], sd, "dep_%d_list_mu.%d." % (k + 1, idx))
                sg[:, :, slots[k][0], slots[k][1]] = _dep_net(deps[k], sd, "dep_%d_list_sigma.%d." % (k + 1, idx))
            _, p = gaussian_conditional_forward(xj, sg, mu, training, n2)                 # :743-744
            sis.append(_neg_log2(p))
            qs.append(q)
        si_list.append(torch.cat(sis, 1))
        con = torch.cat(qs, 1)
        q_list.append(con)
    q_list.reverse()
    si_list.reverse()
    return si_xe, si_list, xe_q, q_list


ENTROPY_LAYERS = {
    "factorized": factorized_forward,
    "conditioned2ZTsepSubbands": conditioned2_forward,
    "onlyEZWT": only_ezwt_forward,
    "DWTConditioned2EntropyLayerZTBlock": ztblock_forward,
}
