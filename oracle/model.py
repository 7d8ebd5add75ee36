"""Oracle: model wrapper, RD loss, colour transforms, agent step -- CPU restatement, test infrastructure only.

Follows (paths relative to /root/reference):
  graphs/models/LiftingBasedDWT_net.py:35-62     LiftingBasedDWTNetWrapper.forward (3 per-plane nets, clrch==1)
  graphs/models/LiftingBasedDWT_net.py:154-170   LiftingBasedDWTNet.forward (encode -> entropymodel -> decode)
  graphs/layers/lifting_dwt_nets.py:724-782      LiftingBasedNeuralWaveletv4.encode / decode
  graphs/layers/lifting_dwt_nets.py:212-277      DWTPytorchWaveletsLayer.encode / decode (CDF 9/7)
  graphs/losses/rate_dist.py:35-42               TrainRDLoss.forward3
  agents/liftingDWT_agent.py:75-111,155-201      per-batch maths of train_one_epoch / validate
  compressai.transforms RGB2YCbCr / YCbCr2RGB (BT.709; package absent -> parity unpinned, closed form)
"""
import torch

from . import cdf97
from .entropy import ENTROPY_LAYERS
from .lifting import lifting_forward, lifting_inverse
from .subband_ae import ae_decode, ae_encode

DEFAULT_CFG = {  # liftingDWT.json:11-23 (hot-path keys)
    "clrch": 1, "netType": "LiftingBasedNeuralWaveletv4", "entropy_layer": "conditioned2ZTsepSubbands",
    "autoencoder": "SubbandAutoEncoder", "dwtlevels": 4, "num_lifting_perlayer": 2, "filtersize": 5,
    "block_property": "same", "scale": 0, "linearity_flag": 1, "depth_scale": 2, "res_connection_weight": 0.1,
    "mode": "test", "imshow_validation": False, "postprocess": "none", "lambda_": 11700,
}


def sub(sd, prefix):
    """View of a flat state dict with ``prefix`` stripped."""
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def encode(x, sd, cfg):
    """autoencoder.encode -> (out_xe, out_xo_list); sd is the autoencoder's state dict."""
    kind = "SubbandAutoEncoder" if cfg["netType"] == "CDF97" else cfg["autoencoder"]  # lifting_dwt_nets.py:236-239
    if cfg["netType"] == "CDF97":
        Yl, Yh = cdf97.dwt_forward(x, cfg["dwtlevels"])
    else:
        Yl, Yh = lifting_forward(x, sd, cfg)
    out_xe = ae_encode(Yl, sd, "Yl_ae.", kind)
    out_xo = []
    for i in range(cfg["dwtlevels"]):
        B, C, T, H, W = Yh[i].shape
        out_xo.append(ae_encode(Yh[i].reshape(B, C * 3, H, W), sd, "Yh_ae.%d." % i, kind))
    return out_xe, out_xo


def decode(out_xe, out_xo_list, sd, cfg):
    kind = "SubbandAutoEncoder" if cfg["netType"] == "CDF97" else cfg["autoencoder"]
    Yl = ae_decode(out_xe, sd, "Yl_ae.", kind)
    Yh = []
    for i in range(cfg["dwtlevels"]):
        d = ae_decode(out_xo_list[i], sd, "Yh_ae.%d." % i, kind)
        B, C, H, W = out_xo_list[i].shape
        Yh.append(d.reshape(B, C // 3, 3, H, W))
    if cfg["netType"] == "CDF97":
        return cdf97.dwt_inverse(Yl, Yh)
    return lifting_inverse(Yl, Yh, sd, cfg)


def net_forward(x, sd, cfg, training=False, noises=None):
    """LiftingBasedDWTNet.forward (LiftingBasedDWT_net.py:154-170); sd has 'autoencoder.' / 'entropymodel.' keys."""
    ae_sd, em_sd = sub(sd, "autoencoder."), sub(sd, "entropymodel.")
    out_xe, out_xo = encode(x, ae_sd, cfg)
    si_xe, si_xo, xe_q, xo_q = ENTROPY_LAYERS[cfg["entropy_layer"]](out_xe, out_xo, em_sd, cfg, training, noises)
    xhat = decode(xe_q, xo_q, ae_sd, cfg)
    return xhat, si_xe, si_xo


def wrapper_forward(x, sd, cfg, training=False, noises=None):
    """LiftingBasedDWTNetWrapper.forward (LiftingBasedDWT_net.py:48-62), clrch==1: model0/1/2 on planes."""
    assert cfg["clrch"] == 1
    xh, se, so = [], [], []
    for c in range(3):
        a, b, l = net_forward(x[:, c:c + 1], sub(sd, "model%d." % c), cfg, training,
                              None if noises is None else noises[c])
        xh.append(a)
        se.append(b)
        so.extend(l)
    return torch.cat(xh, 1), torch.cat(se, 1), so


def rd_loss(x, xhat, si_xe, si_xo_list, lambda_):
    """TrainRDLoss.forward3 (rate_dist.py:35-42) -> (loss, mse, rate1, rate2)."""
    mse = torch.mean((x - xhat) ** 2)
    n = x.numel()
    rate1 = torch.sum(si_xe) / n * 3
    rate2 = 0
    for r in si_xo_list:
        rate2 = rate2 + torch.sum(r) / n * 3
    return rate1 + rate2 + lambda_ * mse, mse, rate1, rate2


# BT.709 (compressai.transforms.functional.rgb2ycbcr / ycbcr2rgb)
_KR, _KG, _KB = 0.2126, 0.7152, 0.0722


def rgb2ycbcr(rgb):
    r, g, b = rgb.chunk(3, -3)
    y = _KR * r + _KG * g + _KB * b
    cb = 0.5 * (b - y) / (1 - _KB) + 0.5
    cr = 0.5 * (r - y) / (1 - _KR) + 0.5
    return torch.cat((y, cb, cr), dim=-3)


def ycbcr2rgb(ycbcr):
    y, cb, cr = ycbcr.chunk(3, -3)
    r = y + (2 - 2 * _KR) * (cr - 0.5)
    b = y + (2 - 2 * _KB) * (cb - 0.5)
    g = (y - _KR * r - _KB * b) / _KG
    return torch.cat((r, g, b), dim=-3)


_YSHIFT = torch.tensor([[[0.5]], [[0.0]], [[0.0]]])


def agent_batch(x, sd, cfg, training=False, clamp=False, noises=None):
    """Per-batch maths of the agent (liftingDWT_agent.py:84-96 train, :171-183 validate), clrch==1.

    x: (B,3,H,W) in [0,1].  Returns dict(xhat, loss, mse, rate1, rate2, psnr, si_xe, si_xo)."""
    y = rgb2ycbcr(x) - _YSHIFT
    yhat, si_xe, si_xo = wrapper_forward(y, sd, cfg, training, noises)
    xhat = ycbcr2rgb(yhat + _YSHIFT) - 0.5
    xs = x - 0.5
    if clamp:
        xhat = xhat.clamp(-0.5, 0.5)
    loss, mse, r1, r2 = rd_loss(xs, xhat, si_xe, si_xo, cfg["lambda_"])
    psnr = 10.0 * torch.log10(1.0 / mse)
    return dict(xhat=xhat, loss=loss, mse=mse, rate1=r1, rate2=r2, psnr=psnr, si_xe=si_xe, si_xo=si_xo, y=y)
