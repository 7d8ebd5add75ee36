"""Oracle: the reference's real-coding loops, per pixel in raster order -- CPU restatement, test infrastructure only.

Follows graphs/models/LiftingBasedDWT_net.py (paths relative to /root/reference):
  :374-456  DWTConditioned2EntropyLayerZTsepSubbands.test      which tensor is coded with which context model
  :458-506  compress_ar                                         k x k crop -> context CNN -> (sigma, mu) at the centre ->
                                                                symbol = round(y - mu), index = build_indexes(sigma),
                                                                y_hat[h, w] = symbol + mu
The range coder itself is oracle/rans.py.  Like the reference this walks every pixel with a CNN call on a tiny crop, so it
is only usable on very small tensors (the product evaluates the same maths as a wavefront on the GPU).
"""
import torch
import torch.nn.functional as F

from . import rans
from .entropy import _csc_stack, masked_conv, upsample2


def compress_ar_symbols(y, sd, csc_prefix, k, scale_table, cgp_prefix=None, param=None):
    """-> (symbols (B,C,H,W) int32, indexes (B,C,H,W) int32, dequantised (B,C,H,W)); batch 1 like the reference."""
    pad = k // 2
    B, C, H, W = y.shape
    y_hat = F.pad(y, (pad, pad, pad, pad)).clone()          # not-yet-coded positions hold the ORIGINAL values (:386,393)
    phat = None if param is None else F.pad(param, (pad, pad, pad, pad))
    sym = torch.zeros(B, C, H, W, dtype=torch.int32)
    idx = torch.zeros(B, C, H, W, dtype=torch.int32)
    for h in range(H):
        for w in range(W):
            crop = y_hat[:, :, h:h + k, w:w + k]
            if cgp_prefix is None:
                ms = _csc_stack(crop, sd, csc_prefix, groups=C)                              # 5 masked 3x3 convs
            else:
                csc = masked_conv(crop, sd, csc_prefix, groups=C)
                p0, p1, p2 = phat[:, :, h:h + k, w:w + k].chunk(3, dim=1)
                c0, c1, c2 = csc.chunk(3, dim=1)
                t = torch.cat((p0, c0, p1, c1, p2, c2), dim=1)
                for n in (0, 2, 4, 6):
                    t = F.conv2d(t, sd[cgp_prefix + "%d.weight" % n], sd[cgp_prefix + "%d.bias" % n], groups=C)
                    if n != 6:
                        t = F.leaky_relu(t, 0.01)
                ms = t
            sigma = ms[:, 0::2, pad, pad]
            mu = ms[:, 1::2, pad, pad]
            yc = crop[:, :, pad, pad]
            q = torch.round(yc - mu)
            y_hat[:, :, h + pad, w + pad] = q + mu
            sym[:, :, h, w] = q.int()
            idx[:, :, h, w] = rans.build_indexes(sigma, scale_table)
    return sym, idx, y_hat[:, :, pad:pad + H, pad:pad + W].contiguous()


def conditioned2_test_symbols(out_xe, out_xo_list, sd, cfg):
    """The encoder half of test() (:374-417) for one plane: -> dict tensor-name -> (symbols, indexes, dequantised);
    names 'xe', 'xo<level>'."""
    L = cfg["dwtlevels"]
    table = rans.get_scale_table()
    out = {"xe": compress_ar_symbols(out_xe, sd, "csc_xe.", 3, table)}
    i = L - 1
    out["xo%d" % i] = compress_ar_symbols(out_xo_list[i], sd, "csc_list.%d." % i, 3, table)
    con = upsample2(out["xo%d" % i][2])
    for i in range(L - 2, -1, -1):
        plc = F.conv2d(con, sd["plc_list.%d.0.weight" % i], sd["plc_list.%d.0.bias" % i], padding=1)
        plc = F.leaky_relu(plc, 0.01)
        plc = F.conv2d(plc, sd["plc_list.%d.2.weight" % i], sd["plc_list.%d.2.bias" % i], padding=1)
        out["xo%d" % i] = compress_ar_symbols(out_xo_list[i], sd, "csc_list.%d." % i, 5, table,
                                              cgp_prefix="cgp_out_xo_list.%d." % i, param=plc)
        con = upsample2(out["xo%d" % i][2])
    return out
