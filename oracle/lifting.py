"""Oracle: learned lifting DWT (forward / inverse) -- CPU restatement, test infrastructure only.

Follows (paths relative to /root/reference):
  graphs/layers/wavelet_forward_v2.py:26-81   one_level_lifting / lifting_forward_row_2_stage_lifting
  graphs/layers/wavelet_inverse_v2.py:20-92   one_level_lifting / reconstruct_fun / lifting_inverse_row_2_stage_lifting
  graphs/layers/P_block_v2.py:40-55           P_block_v2.forward
  graphs/layers/lifting_dwt_nets.py:646-827   LiftingBasedNeuralWaveletv4 (encode :724-746, decode :748-782, skip filters :784-827)

Parameters are passed as a flat dict ``sd`` using the reference's state_dict key
names for one ``LiftingBasedNeuralWaveletv4`` (``P_blocks.{k}.conv{1..4}.{weight,bias}``,
``U_blocks...``, ``preProcessingList.{j}.weight``, ``nh``, ``nl``,
``Yl_ae.*``, ``Yh_ae.{i}.*``).
"""
import torch
import torch.nn.functional as F

# lifting_dwt_nets.py:431-432 (bior4.4 lifting constants + subband gains)
LIFTING_COEFF = [-1.586134342059924, -0.052980118572961, 0.882911075530934,
                 0.443506852043971, 0.869864451624781, 1.149604398860241]


def skip_filter(x, w):
    """3x1 zero-padded conv along dim 2, no bias (lifting_dwt_nets.py:805-819).  w: (1,1,3,1)."""
    return F.conv2d(x, w, None, stride=1, padding=(1, 0))


def p_block(x, sd, prefix, linearity_flag=1):
    """P_block_v2.forward (P_block_v2.py:40-55): conv1 -> tanh -> conv2 -> tanh -> conv3 (+conv1 pre-act) -> conv4."""
    w = lambda n: sd[prefix + n]
    pad = w("conv1.weight").shape[-1] // 2
    out_res = F.conv2d(x, w("conv1.weight"), w("conv1.bias"), padding=pad)
    t = torch.tanh(out_res) if linearity_flag == 1 else out_res
    t = F.conv2d(t, w("conv2.weight"), w("conv2.bias"), padding=pad)
    if linearity_flag == 1:
        t = torch.tanh(t)
    t = F.conv2d(t, w("conv3.weight"), w("conv3.bias"), padding=pad)
    t = t + out_res
    return F.conv2d(t, w("conv4.weight"), w("conv4.bias"), padding=pad)


def _blocks(sd, p_idx):
    return "P_blocks.%d." % p_idx, "U_blocks.%d." % p_idx


def lift_2stage_forward(L, H, sd, cfg, p_off=0):
    """wavelet_forward_v2.py:58-81.  p_off selects the block pair for block_property=='different'."""
    rw = cfg["res_connection_weight"]
    lin = cfg.get("linearity_flag", 1)
    for stage in range(2):
        P, U = _blocks(sd, p_off + stage)
        skip = skip_filter(L, sd["preProcessingList.%d.weight" % (2 * stage)])
        H = H + skip + p_block(skip, sd, P, lin) * rw
        skip = skip_filter(H, sd["preProcessingList.%d.weight" % (2 * stage + 1)])
        L = L + skip + p_block(skip, sd, U, lin) * rw
    if cfg.get("scale", 0) == 1:
        nh = LIFTING_COEFF[4] + sd["nh"] * 0.1
        nl = LIFTING_COEFF[5] + sd["nl"] * 0.1
        H = H * nh
        L = L * nl
    return L, H


def lift_2stage_inverse(L, H, sd, cfg, p_off=0):
    """wavelet_inverse_v2.py:68-92 (exact mirror of the forward, reverse order, minus signs)."""
    rw = cfg["res_connection_weight"]
    lin = cfg.get("linearity_flag", 1)
    if cfg.get("scale", 0) == 1:
        nh = LIFTING_COEFF[4] + sd["nh"] * 0.1
        nl = LIFTING_COEFF[5] + sd["nl"] * 0.1
        H = H / nh
        L = L / nl
    for stage in (1, 0):
        P, U = _blocks(sd, p_off + stage)
        skip = skip_filter(H, sd["preProcessingList.%d.weight" % (2 * stage + 1)])
        L = L - skip - p_block(skip, sd, U, lin) * rw
        skip = skip_filter(L, sd["preProcessingList.%d.weight" % (2 * stage)])
        H = H - skip - p_block(skip, sd, P, lin) * rw
    return L, H


def one_level_forward(x, sd, cfg, p_off=0):
    """wavelet_forward_v2.py:26-54.  Returns (LL, LH, HL, HH); HL = vertical-low / horizontal-high."""
    L = x[:, :, 0::2, :]
    H = x[:, :, 1::2, :]
    L, H = lift_2stage_forward(L, H, sd, cfg, p_off)
    L = torch.transpose(L, 2, 3)
    LL, HL = lift_2stage_forward(L[:, :, 0::2, :], L[:, :, 1::2, :], sd, cfg, p_off)
    LL = torch.transpose(LL, 2, 3)
    HL = torch.transpose(HL, 2, 3)
    H = torch.transpose(H, 2, 3)
    LH, HH = lift_2stage_forward(H[:, :, 0::2, :], H[:, :, 1::2, :], sd, cfg, p_off)
    LH = torch.transpose(LH, 2, 3)
    HH = torch.transpose(HH, 2, 3)
    return LL, LH, HL, HH


def merge(up, bot):
    """wavelet_inverse_v2.py:40-56 reconstruct_fun: interleave along dim 2, then transpose(2,3)."""
    n, c, m, k = up.shape
    recon = torch.empty(n, c, 2 * m, k, dtype=up.dtype)
    recon[:, :, 0::2, :] = up
    recon[:, :, 1::2, :] = bot
    return torch.transpose(recon, 2, 3)


def one_level_inverse(LL, LH, HL, HH, sd, cfg, p_off=0):
    """wavelet_inverse_v2.py:20-38."""
    LL = torch.transpose(LL, 2, 3)
    HL = torch.transpose(HL, 2, 3)
    LL, HL = lift_2stage_inverse(LL, HL, sd, cfg, p_off)
    L = merge(LL, HL)
    LH = torch.transpose(LH, 2, 3)
    HH = torch.transpose(HH, 2, 3)
    LH, HH = lift_2stage_inverse(LH, HH, sd, cfg, p_off)
    H = merge(LH, HH)
    L, H = lift_2stage_inverse(L, H, sd, cfg, p_off)
    recon = merge(L, H)
    return torch.transpose(recon, 2, 3)


def _fwd_off(cfg, level):
    # lifting_dwt_nets.py:711-716: 'different' gives level l the slice [l*ll, (l+1)*ll)
    return 0 if cfg.get("block_property", "same") == "same" else level * cfg.get("num_lifting_perlayer", 2)


def _inv_off(cfg, level):
    # lifting_dwt_nets.py:718-722: the inverse slices all START at waveletLevel*liftingLevel
    if cfg.get("block_property", "same") == "same":
        return 0
    return cfg["dwtlevels"] * cfg.get("num_lifting_perlayer", 2)


def lifting_forward(x, sd, cfg):
    """The transform part of encode (lifting_dwt_nets.py:724-732): returns (LL, [Yh_0..Yh_{L-1}]) finest first,
    Yh_i of shape (B, C, 3, h, w) stacked as (LH, HL, HH)."""
    Yh = []
    LL = x
    for level in range(cfg["dwtlevels"]):
        LL, LH, HL, HH = one_level_forward(LL, sd, cfg, _fwd_off(cfg, level))
        Yh.append(torch.cat((LH.unsqueeze(2), HL.unsqueeze(2), HH.unsqueeze(2)), 2))
    return LL, Yh


def lifting_inverse(Yl, Yh, sd, cfg):
    """The transform part of decode (lifting_dwt_nets.py:762-781)."""
    nlev = cfg["dwtlevels"]
    LL = Yl
    for k in range(nlev):
        lev = nlev - k - 1
        LL = one_level_inverse(LL, Yh[lev][:, :, 0], Yh[lev][:, :, 1], Yh[lev][:, :, 2], sd, cfg, _inv_off(cfg, lev))
    return LL
