"""Tensor-level wrappers over the C-ABI (include/lldwt.h).  PyTorch supplies device memory and the stream only.

Tensor convention: "plane-major" (P, B, C, h, w) fp32 contiguous CUDA(HIP) tensors; P = number of per-plane networks
(3 for clrch == 1), parameters stacked on a leading P axis.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, EPI_LRELU_BWD, EPI_NONE, EPI_TANH_BWD, ConvDesc, View,  # noqa: F401
                   check)

_ws = {}


def plc_mode():
    """Arithmetic of the dense 243 -> 243 3x3 tree-context conv (LiftingBasedDWT_net.py:271-272): 'f32' = fp32 MFMA
    (reference arithmetic), 'f16x3' = split-fp16 (power-of-two scaled hi*hi + hi*lo + lo*hi on the fp16 matrix cores, fp32
    accumulate; ~2^-21 relative per product, csrc/conv_f16x3.hip).
    Default 'f16x3' (parity-gated by tests/test_gpu_fullsize_oracle.py at the same bars as 'f32'); the fp32 kernel
    stays available as the exact-arithmetic fallback.  Environment variable LLDWT_PLC_MODE; read on every call so tests
    can switch it."""
    import os
    m = os.environ.get("LLDWT_PLC_MODE", "f16x3")
    if m not in ("f32", "f16x3"):
        raise _lib.LLDWTError("LLDWT_PLC_MODE must be 'f32' or 'f16x3' (got %r)" % m)
    return m


def train_lift_f16():
    """True when the training forward of the lifting steps runs on the fused f16x3 kernel (lldwt_train_lift_f16: lift mode f16x3
    and LLDWT_TRAIN_LIFT != f32); the packed P/U blocks must then carry their split-fp16 section."""
    return bool(_lib.load().lldwt_train_lift_f16())


def set_precision(name):
    """Arithmetic of the eval path's matrix kernels (fused lifting step, tree-context pair, cgp chain): 'f16x3' (default: three
    fp16 MFMA products per fp32 MAC, fp32-level accuracy), 'fp16' or 'bf16' (ONE product per MAC on operands rounded to that
    type, fp32 accumulate; BASELINE configs[4] / configs[1], tolerance class 1e-2 -- never the headline).  Also settable with
    the environment variable LLDWT_PRECISION before the library is loaded.  Training is not affected."""
    if name not in _lib.PRECISIONS:
        raise _lib.LLDWTError("precision must be one of %s (got %r)" % (sorted(_lib.PRECISIONS), name))
    check(_lib.load().lldwt_set_precision(_lib.PRECISIONS[name]), "lldwt_set_precision")


def get_precision():
    code = _lib.load().lldwt_get_precision()
    return [k for k, v in _lib.PRECISIONS.items() if v == code][0]


_diag_keep = {}


def set_diagnostics(kind, stamps=None, flags=0):
    """Diagnostics hook of the split-fp16 kernels (lldwt_set_diagnostics; tools/*_stamps.py and the composed-vs-sequential
    tests only): kind 0 = fused lifting step (flags = its debug mask), 1 = tree-pair conv, 2 = cgp chain.  ``stamps``: an
    int64 device tensor the kernels write s_memtime stamps into (kept alive here until it is replaced), None = off."""
    _diag_keep[kind] = stamps
    ptr = C.c_void_p(stamps.data_ptr()) if stamps is not None else C.c_void_p(0)
    nbytes = stamps.numel() * stamps.element_size() if stamps is not None else 0
    check(_lib.load().lldwt_set_diagnostics(int(kind), ptr, nbytes, int(flags)), "lldwt_set_diagnostics")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise _lib.LLDWTError("%s must be a contiguous fp32 device tensor (got %s)" % (
            name, None if t is None else (t.device, t.dtype, t.is_contiguous())))
    return C.c_void_p(t.data_ptr())


def _opt(t, name="tensor"):
    return C.c_void_p(0) if t is None else _chk(t, name)


def workspace(nbytes, device):
    """Grow-only scratch buffer per device (the library never allocates: include/lldwt.h conventions)."""
    key = (device.type, device.index)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def rgb_to_ycc(x):
    """(B,3,H,W) RGB in [0,1] -> plane-major (3,B,1,H,W) YCbCr with Y-0.5 (agents/liftingDWT_agent.py:86-87)."""
    lib = _lib.load()
    B, c, H, W = x.shape
    assert c == 3
    y = torch.empty(3, B, 1, H, W, device=x.device, dtype=torch.float32)
    check(lib.lldwt_rgb_to_ycc(_chk(x, "x"), _chk(y), B, H, W, _stream()), "rgb_to_ycc")
    return y


def u8hwc_to_f32chw(src):
    """(B,H,W,3) uint8 device tensor -> (B,3,H,W) fp32 in [0,1] (ToTensor semantics, dataloaders/image_dl.py:81)."""
    if not (isinstance(src, torch.Tensor) and src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
            and src.dim() == 4 and src.shape[3] == 3):
        raise _lib.LLDWTError("u8hwc_to_f32chw: expected a contiguous (B,H,W,3) uint8 device tensor")
    B, H, W, _ = src.shape
    dst = torch.empty(B, 3, H, W, device=src.device, dtype=torch.float32)
    check(_lib.load().lldwt_u8hwc_to_f32chw(C.c_void_p(src.data_ptr()), _chk(dst), B, H, W, _stream()), "u8hwc_to_f32chw")
    return dst


def ycc_to_rgb(y, clamp=False):
    """plane-major (3,B,1,H,W) -> (B,3,H,W) RGB-0.5 (agents/liftingDWT_agent.py:90-94, clamp :181)."""
    lib = _lib.load()
    _, B, _, H, W = y.shape
    x = torch.empty(B, 3, H, W, device=y.device, dtype=torch.float32)
    check(lib.lldwt_ycc_to_rgb(_chk(y, "y"), _chk(x), B, H, W, int(bool(clamp)), _stream()), "ycc_to_rgb")
    return x


def pblock_packed_floats(Cc, K):
    return int(_lib.load().lldwt_pblock_packed_floats(Cc, K))


def pack_pblock(w1, b1, w2, b2, w3, b3, w4, b4, train=False, compose=True):
    """Stacked P_block_v2 parameters (planes, ...) in PyTorch layout -> packed (planes, total) buffer.  train=True: for the
    fp32 training kernels only (the split-fp16 section of the buffer is not written).  compose=False: the split-fp16 section
    without the composed 9x9 kernels of the eval path (lldwt_pack_pblock_seq: the fused training forward's pack)."""
    lib = _lib.load()
    planes, Cc, _, K, _ = w1.shape
    out = torch.empty(planes, pblock_packed_floats(Cc, K), device=w1.device, dtype=torch.float32)
    fn = lib.lldwt_pack_pblock_train if train else (lib.lldwt_pack_pblock if compose else lib.lldwt_pack_pblock_seq)
    check(fn(_chk(w1), _chk(b1), _chk(w2), _chk(b2), _chk(w3), _chk(b3), _chk(w4), _chk(b4), _chk(out), planes, Cc, K,
             _stream()), "pack_pblock")
    return out


def bwd_lift_f16():
    """True when lift_step_bwd with a backward pack runs the fused split-fp16 backward-data launch (include/lldwt.h)."""
    return bool(_lib.load().lldwt_bwd_lift_f16())


def pack_pblock_bwd(w1, w2, w3, w4):
    """Backward pack of a tanh P_block_v2 (planes, 16, ., 5, 5): transposed, mirrored weights in the forward pack's layout."""
    lib = _lib.load()
    planes, Cc, _, K, _ = w1.shape
    out = torch.empty(planes, pblock_packed_floats(Cc, K), device=w1.device, dtype=torch.float32)
    nb = lib.lldwt_pack_pblock_bwd_ws_bytes(planes)
    ws = workspace(nb, w1.device)
    check(lib.lldwt_pack_pblock_bwd(_chk(w1), _chk(w2), _chk(w3), _chk(w4), _chk(out), C.c_void_p(ws.data_ptr()), nb, planes,
                                    Cc, K, _stream()), "pack_pblock_bwd")
    return out


def view_of(t, z, h, w, offset=0, sz=None, sy=None, sx=1):
    """lldwt_view over the storage of ``t`` (element offsets/strides)."""
    return View(C.c_void_p(t.data_ptr() + 4 * offset), sz if sz is not None else h * w, sy if sy is not None else w, sx)


def lift_step(src, dst_in, dst_out, Z, batch, h, w, taps, packed, Cc, K, vertical, sign, res_weight, linear=False):
    """One lifting step on lldwt_view arguments (see include/lldwt.h)."""
    lib = _lib.load()
    nb = lib.lldwt_lift_step_ws_bytes(Z, h, w, Cc)
    ws = workspace(nb, taps.device)
    check(lib.lldwt_lift_step(src, dst_in, dst_out, Z, batch, h, w, _chk(taps), _chk(packed), Cc, K, int(vertical),
                              float(sign), float(res_weight), int(bool(linear)), C.c_void_p(ws.data_ptr()), nb,
                              _stream()), "lift_step")


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def lifting_forward(x, taps, packed, levels, Cc, K, res_weight, linear=False, different=False, scale_nh=None,
                    scale_nl=None, block_offset=0):
    """x: (P,B,1,H,W) -> (ll (P,B,1,H>>L,W>>L), [yh_i (P,B,3,H>>(i+1),W>>(i+1))])  (lifting_dwt_nets.py:728-732)."""
    lib = _lib.load()
    P, B, _, H, W = x.shape
    dev = x.device
    ll = torch.empty(P, B, 1, H >> levels, W >> levels, device=dev, dtype=torch.float32)
    yh = [torch.empty(P, B, 3, H >> (i + 1), W >> (i + 1), device=dev, dtype=torch.float32) for i in range(levels)]
    nb = lib.lldwt_lifting_ws_bytes(P * B, H, W, Cc)
    ws = workspace(nb, dev)
    check(lib.lldwt_lifting_forward(_chk(x, "x"), _chk(ll), _ptr_array(yh), P, B, H, W, levels, _chk(taps, "taps"),
                                    _chk(packed, "packed"), int(packed.shape[1]), int(block_offset),
                                    int(bool(different)), Cc, K, float(res_weight), int(bool(linear)), _opt(scale_nh),
                                    _opt(scale_nl), C.c_void_p(ws.data_ptr()), nb, _stream()), "lifting_forward")
    return ll, yh


def lifting_inverse(ll, yh, taps, packed, Cc, K, res_weight, linear=False, scale_nh=None, scale_nl=None,
                    block_offset=0):
    lib = _lib.load()
    levels = len(yh)
    P, B, _, hl, wl = ll.shape
    H, W = hl << levels, wl << levels
    dev = ll.device
    x = torch.empty(P, B, 1, H, W, device=dev, dtype=torch.float32)
    for t in yh:
        _chk(t, "yh")
    nb = lib.lldwt_lifting_ws_bytes(P * B, H, W, Cc)
    ws = workspace(nb, dev)
    check(lib.lldwt_lifting_inverse(_chk(ll, "ll"), _ptr_array(yh), _chk(x), P, B, H, W, levels, _chk(taps, "taps"),
                                    _chk(packed, "packed"), int(packed.shape[1]), int(block_offset), Cc, K,
                                    float(res_weight), int(bool(linear)), _opt(scale_nh), _opt(scale_nl),
                                    C.c_void_p(ws.data_ptr()), nb, _stream()), "lifting_inverse")
    return x


def set_cdf97_short_levels(periodic):
    """Policy for CDF 9/7 level inputs shorter than the 10-tap filter (lldwt_set_cdf97_short_levels): False (default) =
    such a call raises LLDWTError (the reference's single-fold form and the exact periodic transform differ there),
    True = compute the periodic form (= PyWavelets)."""
    check(_lib.load().lldwt_set_cdf97_short_levels(1 if periodic else 0), "lldwt_set_cdf97_short_levels")


def cdf97_forward(x, levels, adj=False):
    """x: (Z..., H, W) as (P,B,C,H,W) -> (ll, [yh_i (P,B,C*3? no: (P,B*C,3,h,w))]).  Channels are folded into batch."""
    lib = _lib.load()
    P, B, Cc, H, W = x.shape
    Z = P * B * Cc
    dev = x.device
    ll = torch.empty(P, B, Cc, H >> levels, W >> levels, device=dev, dtype=torch.float32)
    yh = [torch.empty(P, B, Cc, 3, H >> (i + 1), W >> (i + 1), device=dev, dtype=torch.float32) for i in range(levels)]
    nb = lib.lldwt_cdf97_ws_bytes(Z, H, W)
    ws = workspace(nb, dev)
    check(lib.lldwt_cdf97_forward_ex(_chk(x, "x"), _chk(ll), _ptr_array(yh), Z, H, W, levels, int(bool(adj)),
                                     C.c_void_p(ws.data_ptr()), nb, _stream()), "cdf97_forward")
    return ll, yh


def cdf97_inverse(ll, yh, adj=False):
    lib = _lib.load()
    levels = len(yh)
    P, B, Cc, hl, wl = ll.shape
    H, W = hl << levels, wl << levels
    Z = P * B * Cc
    x = torch.empty(P, B, Cc, H, W, device=ll.device, dtype=torch.float32)
    for t in yh:
        _chk(t, "yh")
    nb = lib.lldwt_cdf97_ws_bytes(Z, H, W)
    ws = workspace(nb, ll.device)
    check(lib.lldwt_cdf97_inverse_ex(_chk(ll, "ll"), _ptr_array(yh), _chk(x), Z, H, W, levels, int(bool(adj)),
                                     C.c_void_p(ws.data_ptr()), nb, _stream()), "cdf97_inverse")
    return x


def subband_mlp(x, w0, b0, w1, b1, w2, b2, w3, b3, transposed=False, hidden=32):
    """SubbandAutoEncoder encode/decode (lifting_dwt_nets.py:99-110); x: (P,B,C,h,w), weights stacked (P,...)."""
    lib = _lib.load()
    P, B, Cc, h, w = x.shape
    y = torch.empty_like(x)
    check(lib.lldwt_subband_mlp(_chk(x, "x"), _chk(y), P, B, Cc, h * w, hidden, _chk(w0), _chk(b0), _chk(w1), _chk(b1),
                                _chk(w2), _chk(b2), _chk(w3), _chk(b3), int(bool(transposed)), _stream()), "subband_mlp")
    return y


def subband_mlp_bwd(x, gy, w0, b0, w1, b1, w2, b2, w3, hidden=32):
    """Backward-data of the encode-layout subband MLP: -> (gx, [h0,h1,h2], [d0,d1,d2]); hidden tensors (P,B,C*hidden,h,w)."""
    P, B, Cc, h, w = x.shape
    gx = torch.empty_like(x)
    hs = [torch.empty(P, B, Cc * hidden, h, w, device=x.device, dtype=torch.float32) for _ in range(6)]
    check(_lib.load().lldwt_subband_mlp_bwd(_chk(x, "x"), _chk(gy, "gy"), _chk(gx), *[_chk(t) for t in hs], P, B, Cc, h * w,
                                            hidden, _chk(w0), _chk(b0), _chk(w1), _chk(b1), _chk(w2), _chk(b2), _chk(w3),
                                            _stream()), "subband_mlp_bwd")
    return gx, hs[:3], hs[3:]


def subband_mlp_bwd_w(x, gy, w0, b0, w1, b1, w2, b2, w3, hidden=32):
    """Backward of the encode-layout subband MLP with the parameter gradients formed in the kernel (lldwt_subband_mlp_bwd_w):
    -> (gx, [dw0, db0, dw1, db1, dw2, db2, dw3, db3]) in the shapes of the parameters."""
    lib = _lib.load()
    P, B, Cc, h, w = x.shape
    gx = torch.empty_like(x)
    shapes = [(P, Cc * hidden, 1, 1, 1), (P, Cc * hidden), (P, Cc * hidden, hidden, 1, 1), (P, Cc * hidden),
              (P, Cc * hidden, hidden, 1, 1), (P, Cc * hidden), (P, Cc, hidden, 1, 1), (P, Cc)]
    g = [torch.empty(s, device=x.device, dtype=torch.float32) for s in shapes]
    nb = lib.lldwt_subband_mlp_bwd_w_ws_bytes(P, Cc, h * w)
    ws = workspace(nb, x.device)
    check(lib.lldwt_subband_mlp_bwd_w(_chk(x, "x"), _chk(gy, "gy"), _chk(gx), P, B, Cc, h * w, hidden, _chk(w0), _chk(b0),
                                      _chk(w1), _chk(b1), _chk(w2), _chk(b2), _chk(w3), *[_chk(t) for t in g],
                                      C.c_void_p(ws.data_ptr()), nb, _stream()), "subband_mlp_bwd_w")
    return gx, g


def conv_desc(cin, cout, K, groups=1, act=ACT_NONE, upsample2=False, transposed=False, tap_mask=None, oc_block=None,
              oc_stride=0, oc_off=0, ytot=None, ic_block=0, ic_stride=0, ic_off=0, xtot=0, epi=0):
    return ConvDesc(cin, cout, K, groups, act, int(bool(upsample2)), int(bool(transposed)),
                    (1 << (K * K)) - 1 if tap_mask is None else int(tap_mask), cout if oc_block is None else oc_block,
                    oc_stride, oc_off, cout if ytot is None else ytot, ic_block, ic_stride, ic_off, xtot, epi)


def flip_mask(mask, K):
    """Tap mask of the 180-degree rotated kernel (backward-data of a masked conv)."""
    if mask is None:
        return None
    KK = K * K
    return sum(1 << (KK - 1 - t) for t in range(KK) if (mask >> t) & 1)


def conv_pack(w, K, groups=1, transposed=False, tap_mask=None, swap_hw=False):
    """(P,cout,cin/groups,K,K) -> packed (P, floats) in MFMA A-operand order.  transposed: the weight is in
    ConvTranspose2d layout (P,cin,cout/groups,K,K) (or a forward Conv2d weight used for its backward-data pass:
    then `cin` is the forward cout) and the taps are flipped; tap_mask refers to the effective (flipped) taps."""
    lib = _lib.load()
    P = w.shape[0]
    if transposed:
        cin, cout = w.shape[1], w.shape[2] * groups
    else:
        cout, cin = w.shape[1], w.shape[2] * groups
    d = conv_desc(cin, cout, K, groups, transposed=transposed, tap_mask=tap_mask)
    n = lib.lldwt_conv_packed_floats(C.byref(d))
    packed = torch.empty(P, n, device=w.device, dtype=torch.float32)
    check(lib.lldwt_conv_pack_ex(_chk(w, "w"), _chk(packed), C.byref(d), P, int(bool(swap_hw)), _stream()), "conv_pack")
    return packed


def conv2d(x, w, bias, K, groups=1, act=ACT_NONE, upsample2=False, transposed=False, tap_mask=None, out=None,
           oc_block=None, oc_stride=0, oc_off=0, direct=False, packed=None, residual=None, aux=None, epi=0,
           ic_block=0, ic_stride=0, ic_off=0, cin=None, absmax=None):
    """General conv layer (include/lldwt.h lldwt_conv2d).  x: (P,B,xtot,h,w); w: (P,cout,cin/groups,K,K)
    (transposed: (P,cin,cout/groups,K,K)).  ``out``: optional pre-allocated (P,B,ytot,h,w) tensor for channel placement.
    ``packed``: result of conv_pack(w, ...) to skip re-packing; ``direct``: reference-order VALU kernel;
    ``absmax``: optional (P,64) fp32 tensor that receives max|y| per plane (lldwt_conv2d_absmax)."""
    lib = _lib.load()
    P, B, xtot, hi, wi = x.shape
    if transposed:
        cin_w, cout = w.shape[1], w.shape[2] * groups
    else:
        cout, cin_w = w.shape[1], w.shape[2] * groups
    cin = cin_w if cin is None else cin
    h, wd = (hi * 2, wi * 2) if upsample2 else (hi, wi)
    if out is None:
        out = torch.empty(P, B, cout, h, wd, device=x.device, dtype=torch.float32)
    d = conv_desc(cin, cout, K, groups, act, upsample2, transposed, tap_mask, oc_block, oc_stride, oc_off, out.shape[2],
                  ic_block, ic_stride, ic_off, xtot if ic_block else 0, epi)
    if not ic_block and xtot != cin:
        raise _lib.LLDWTError("conv2d: input has %d channels, weight expects %d" % (xtot, cin))
    if direct:
        assert residual is None and aux is None and not ic_block
        check(lib.lldwt_conv2d_direct(_chk(x, "x"), _chk(out, "out"), _chk(w, "w"), _opt(bias, "bias"), C.byref(d), P, B,
                                      h, wd, _stream()), "conv2d_direct")
        return out
    if packed is None:
        packed = conv_pack(w, K, groups, transposed, tap_mask)
    if absmax is not None:
        check(lib.lldwt_conv2d_absmax(_chk(x, "x"), _chk(out, "out"), _chk(packed, "packed"), _opt(bias, "bias"),
                                      _opt(residual, "residual"), _opt(aux, "aux"), _chk(absmax, "absmax"), C.byref(d), P, B,
                                      h, wd, _stream()), "conv2d_absmax")
        return out
    check(lib.lldwt_conv2d(_chk(x, "x"), _chk(out, "out"), _chk(packed, "packed"), _opt(bias, "bias"),
                           _opt(residual, "residual"), _opt(aux, "aux"), C.byref(d), P, B, h, wd, _stream()), "conv2d")
    return out


def storage_dtype():
    """Storage type of the tree-context tensor between the two tree convs: 'fp32' (default, the reference's) or 'fp16'
    (BASELINE configs[4]: half the bytes, two MFMA products instead of three, 1e-2 tolerance class).  LLDWT_STORAGE."""
    import os
    m = os.environ.get("LLDWT_STORAGE", "fp32")
    if m not in ("fp32", "fp16"):
        raise _lib.LLDWTError("LLDWT_STORAGE must be 'fp32' or 'fp16' (got %r)" % m)
    return m


def conv2d_f16out(x, w, bias, K, oscale, act=ACT_NONE, upsample2=False, packed=None):
    """conv2d whose output is STORED as fp16 x oscale[plane] (include/lldwt.h lldwt_conv2d_f16out) -> (P,B,cout,h,w) half."""
    lib = _lib.load()
    P, B, cin, hi, wi = x.shape
    cout = w.shape[1]
    h, wd = (hi * 2, wi * 2) if upsample2 else (hi, wi)
    y = torch.empty(P, B, cout, h, wd, device=x.device, dtype=torch.float16)
    d = conv_desc(cin, cout, K, 1, act, upsample2, False, None, None, 0, 0, cout, 0, 0, 0, 0, 0)
    if packed is None:
        packed = conv_pack(w, K)
    check(lib.lldwt_conv2d_f16out(_chk(x, "x"), C.c_void_p(y.data_ptr()), _chk(packed, "packed"), _opt(bias, "bias"),
                                  _chk(oscale, "oscale"), C.byref(d), P, B, h, wd, _stream()), "conv2d_f16out")
    return y


def conv3x3_f16in(x16, packed, bias, cout, xscale, act=ACT_NONE):
    """Dense 3x3 conv reading an fp16-stored input (lldwt_conv3x3_f16in); packed = conv_f16x3_pack(w)."""
    if not (x16.is_cuda and x16.dtype == torch.float16 and x16.is_contiguous()):
        raise _lib.LLDWTError("conv3x3_f16in: x16 must be a contiguous fp16 device tensor")
    P, B, cin, h, w = x16.shape
    y = torch.empty(P, B, cout, h, w, device=x16.device, dtype=torch.float32)
    check(_lib.load().lldwt_conv3x3_f16in(C.c_void_p(x16.data_ptr()), _chk(y), C.c_void_p(packed.data_ptr()), _opt(bias, "bias"),
                                         _chk(xscale, "xscale"), cin, cout, act, P, B, h, w, _stream()), "conv3x3_f16in")
    return y


def plc_fuse():
    """Whether the eval path computes the tree-context PAIR in one launch (lldwt_plc_fused: the first conv on the fly
    inside the second's staging, no 243-channel tensor in HBM).  Only with plc_mode() == 'f16x3' and fp32 storage.
    Environment variable LLDWT_PLC_FUSE (default 1)."""
    import os
    v = os.environ.get("LLDWT_PLC_FUSE", "1")
    if v not in ("0", "1"):
        raise _lib.LLDWTError("LLDWT_PLC_FUSE must be 0 or 1 (got %r)" % v)
    return v == "1"


def plc_fused_pack1(w1, b1):
    """(P,cmid,3,3,3), (P,cmid) fp32 -> packed split-fp16 first tree conv for plc_fused (uint8 (P, bytes))."""
    lib = _lib.load()
    P, cmid, cin, K, K2 = w1.shape
    if cin != 3 or K != 3 or K2 != 3:
        raise _lib.LLDWTError("plc_fused_pack1: a (P,cmid,3,3,3) weight")
    nb = int(lib.lldwt_plc_fused_pack1_bytes(cmid))
    if nb <= 0:
        raise _lib.LLDWTError("plc_fused_pack1: cmid must be in 1..256")
    packed = torch.empty(P, nb, device=w1.device, dtype=torch.uint8)
    check(lib.lldwt_plc_fused_pack1(_chk(w1, "w1"), _chk(b1, "b1"), C.c_void_p(packed.data_ptr()), cmid, P, _stream()),
          "plc_fused_pack1")
    return packed


def plc_fused(parent, packed1, packed2, bias2, cmid, cout, act=ACT_NONE):
    """y = act(conv3x3(LeakyReLU(conv3x3(up2(parent)) + b1)) + b2) in one launch (include/lldwt.h lldwt_plc_fused)."""
    P, B, cin, hp, wp = parent.shape
    if cin != 3:
        raise _lib.LLDWTError("plc_fused: the parent has 3 channels")
    y = torch.empty(P, B, cout, 2 * hp, 2 * wp, device=parent.device, dtype=torch.float32)
    check(_lib.load().lldwt_plc_fused(_chk(parent, "parent"), _chk(y), C.c_void_p(packed1.data_ptr()),
                                     C.c_void_p(packed2.data_ptr()), _opt(bias2, "bias2"), cmid, cout, act, P, B,
                                     2 * hp, 2 * wp, _stream()), "plc_fused")
    return y


def cgp_mode():
    """Arithmetic of the fused cgp stack on the eval path: 'f16x3' (default; split-fp16 register chain, csrc/cgp_f16x3.hip)
    or 'f32' (fp32 MFMA kernel k_cgp_rate).  Environment variable LLDWT_CGP_MODE."""
    import os
    m = os.environ.get("LLDWT_CGP_MODE", "f16x3")
    if m not in ("f32", "f16x3"):
        raise _lib.LLDWTError("LLDWT_CGP_MODE must be 'f32' or 'f16x3' (got %r)" % m)
    return m


def cgp16_supported(ws, groups):
    c = [ws[0].shape[2]] + [w.shape[1] // groups for w in ws]
    return _lib.load().lldwt_cgp16_packed_bytes(c[0], c[1], c[2], c[3], groups) > 0 and c[4] == 2


def cgp16_pack(ws, bs, groups):
    """The four (folded) 1x1 weights (P, groups*c_{l+1}, c_l, 1, 1) + biases -> packed split-fp16 fragments (uint8 (P, bytes))."""
    lib = _lib.load()
    P = ws[0].shape[0]
    c = [ws[0].shape[2]] + [w.shape[1] // groups for w in ws]
    nb = int(lib.lldwt_cgp16_packed_bytes(c[0], c[1], c[2], c[3], groups))
    if nb <= 0:
        raise _lib.LLDWTError("cgp16_pack: dimensions %s not built (93 -> 162 -> 54 -> 18 -> 2 only)" % (c,))
    packed = torch.empty(P, nb, device=ws[0].device, dtype=torch.uint8)
    args = []
    for w, b in zip(ws, bs):
        args += [_chk(w, "w"), _chk(b, "b")]
    check(lib.lldwt_cgp16_pack(*args, C.c_void_p(packed.data_ptr()), P, c[0], c[1], c[2], c[3], groups, _stream()), "cgp16_pack")
    return packed


def cgp16_params(plc, xq, packed16, K, tap_mask):
    """(sigma, mu) of every coefficient from the tree-context features plc (P,B,G*81,h,w) and the quantised subbands xq
    (P,B,G,h,w): -> params (P,B,2G,h,w) (include/lldwt.h lldwt_cgp16_params)."""
    P, B, G, h, w = xq.shape
    params = torch.empty(P, B, 2 * G, h, w, device=xq.device, dtype=torch.float32)
    check(_lib.load().lldwt_cgp16_params(_chk(plc, "plc"), _chk(xq, "xq"), C.c_void_p(packed16.data_ptr()), _chk(params), P, B,
                                        h, w, G, K, int(tap_mask), _stream()), "cgp16_params")
    return params


def cgp16_params_train(plc, xq, packed16, K, tap_mask):
    """Training forward of the cgp stack on the split-fp16 register chain (lldwt_cgp16_params_train):
    -> (params (P,B,2G,h,w), h1 (P,B,G*162,h,w), h2 (P,B,G*54,h,w), h3 (P,B,G*18,h,w))."""
    P, B, G, h, w = xq.shape
    params = torch.empty(P, B, 2 * G, h, w, device=xq.device, dtype=torch.float32)
    hs = [torch.empty(P, B, G * c, h, w, device=xq.device, dtype=torch.float32) for c in (162, 54, 18)]
    check(_lib.load().lldwt_cgp16_params_train(_chk(plc, "plc"), _chk(xq, "xq"), C.c_void_p(packed16.data_ptr()), _chk(params),
                                              _chk(hs[0]), _chk(hs[1]), _chk(hs[2]), P, B, h, w, G, K, int(tap_mask), _stream()),
          "cgp16_params_train")
    return params, hs[0], hs[1], hs[2]


def cgp16_pack_bwd(ws, groups):
    """The four forward 1x1 weights (P, groups*c_{l+1}, c_l, 1, 1) -> the transposed split-fp16 pack of cgp16_bwd (uint8 (P, bytes))."""
    lib = _lib.load()
    P = ws[0].shape[0]
    c = [ws[0].shape[2]] + [w.shape[1] // groups for w in ws]
    nb = int(lib.lldwt_cgp16_bwd_packed_bytes(c[0], c[1], c[2], c[3], groups))
    if nb <= 0:
        raise _lib.LLDWTError("cgp16_pack_bwd: dimensions %s not built (93 -> 162 -> 54 -> 18 -> 2 only)" % (c,))
    packed = torch.empty(P, nb, device=ws[0].device, dtype=torch.uint8)
    check(lib.lldwt_cgp16_pack_bwd(*[_chk(w, "w") for w in ws], C.c_void_p(packed.data_ptr()), P, c[0], c[1], c[2], c[3], groups,
                                   _stream()), "cgp16_pack_bwd")
    return packed


def cgp16_bwd(dparams, h1, h2, h3, packed_bwd16, groups):
    """Backward-data of the cgp stack on the split-fp16 register chain (lldwt_cgp16_bwd)
    -> (dplc (P,B,G*81,h,w), dtaps (P,B,G*12,h,w), d1, d2, d3)."""
    P, B, _, h, w = dparams.shape
    d1, d2, d3 = torch.empty_like(h1), torch.empty_like(h2), torch.empty_like(h3)
    dplc = torch.empty(P, B, groups * 81, h, w, device=dparams.device, dtype=torch.float32)
    dtaps = torch.empty(P, B, groups * 12, h, w, device=dparams.device, dtype=torch.float32)
    check(_lib.load().lldwt_cgp16_bwd(_chk(dparams), _chk(h1), _chk(h2), _chk(h3), C.c_void_p(packed_bwd16.data_ptr()), _chk(d1),
                                      _chk(d2), _chk(d3), _chk(dplc), _chk(dtaps), P, B, h * w, groups, _stream()), "cgp16_bwd")
    return dplc, dtaps, d1, d2, d3


def plc_shape():
    """MFMA shape of the split-fp16 3x3 conv kernels in this process: 32 (32x32x16, default) or 16 (LLDWT_PLC_SHAPE=16)."""
    return 16 if _lib.load().lldwt_plc_shape16() else 32


def conv_f16x3_pack(w):
    """(P,cout,cin,3,3) fp32 -> packed split-fp16 weights (uint8 tensor (P, bytes)) for conv3x3_f16x3."""
    lib = _lib.load()
    P, cout, cin, K, K2 = w.shape
    if K != 3 or K2 != 3:
        raise _lib.LLDWTError("conv_f16x3_pack: 3x3 kernels only")
    nb = int(lib.lldwt_conv_f16x3_packed_bytes(cin, cout))
    packed = torch.empty(P, nb, device=w.device, dtype=torch.uint8)
    check(lib.lldwt_conv_f16x3_pack(_chk(w, "w"), C.c_void_p(packed.data_ptr()), cin, cout, P, _stream()), "conv_f16x3_pack")
    return packed


def absmax_slots(x):
    """x (P, ...) -> slots (P,64) fp32 whose maximum per plane is max|x[p]| (device-side; feeds conv3x3_f16x3)."""
    P = x.shape[0]
    slots = torch.empty(P, 64, device=x.device, dtype=torch.float32)
    check(_lib.load().lldwt_absmax_slots(_chk(x, "x"), P, x.numel() // P, _chk(slots), _stream()), "absmax_slots")
    return slots


def conv3x3_f16x3(x, packed, bias, cout, act=ACT_NONE, slots=None):
    """Dense 3x3 conv on the fp16 matrix cores with split-fp16 operands (include/lldwt.h lldwt_conv3x3_f16x3)."""
    P, B, cin, h, w = x.shape
    if slots is None:
        slots = absmax_slots(x)
    y = torch.empty(P, B, cout, h, w, device=x.device, dtype=torch.float32)
    check(_lib.load().lldwt_conv3x3_f16x3(_chk(x, "x"), _chk(y), C.c_void_p(packed.data_ptr()), _opt(bias, "bias"),
                                         _chk(slots, "slots"), cin, cout, act, P, B, h, w, _stream()), "conv3x3_f16x3")
    return y


def conv2d_wgrad(x, dy, wshape, K, groups=1, upsample2=False, tap_mask=None, want_bias=True, oc_block=None, oc_stride=0,
                 oc_off=0, ic_block=0, ic_stride=0, ic_off=0, dw=None, db=None, alpha=1.0, swap_hw=False):
    """-> (dw (P,cout,cin/groups,K,K), dbias (P,cout) or None); dy: (P,B,ytot,h,w) read through the output placement."""
    lib = _lib.load()
    P, B, ytot, h, wd = dy.shape
    cout, cin = wshape[1], wshape[2] * groups
    if dw is None:
        dw = torch.zeros(wshape, device=x.device, dtype=torch.float32)
    if db is None and want_bias:
        db = torch.zeros(P, cout, device=x.device, dtype=torch.float32)
    d = conv_desc(cin, cout, K, groups, 0, upsample2, False, tap_mask, oc_block, oc_stride, oc_off, ytot, ic_block,
                  ic_stride, ic_off, x.shape[2] if ic_block else 0, 0)
    check(lib.lldwt_conv2d_wgrad_ex(_chk(x, "x"), _chk(dy, "dy"), _chk(dw), _opt(db), C.byref(d), P, B, h, wd,
                                    float(alpha), int(bool(swap_hw)), _stream()), "conv2d_wgrad")
    return dw, db


def wgrad16_f16x3(x, dy, dw=None, db=None, alpha=1.0, swap_hw=False):
    """Backward-weights of a 16 -> 16 5x5 conv whose input is bounded by 1 (tanh outputs) on the fp16 matrix cores, split-fp16
    operands (lldwt_wgrad16_f16x3).  x, dy (P,B,16,h,w) -> (dw (P,16,16,5,5), db (P,16)), accumulated into dw / db if given."""
    P, B, Cc, h, wd = x.shape
    if Cc != 16 or dy.shape != x.shape:
        raise _lib.LLDWTError("wgrad16_f16x3: shapes %r %r" % (tuple(x.shape), tuple(dy.shape)))
    if dw is None:
        dw = torch.zeros(P, 16, 16, 5, 5, device=x.device, dtype=torch.float32)
    if db is None:
        db = torch.zeros(P, 16, device=x.device, dtype=torch.float32)
    slots = workspace(P * 64 * 4, x.device)
    check(_lib.load().lldwt_wgrad16_f16x3(_chk(x, "x"), _chk(dy, "dy"), _chk(dw), _chk(db), C.c_void_p(slots.data_ptr()), P, B, h, wd,
                                          float(alpha), int(bool(swap_hw)), _stream()), "wgrad16_f16x3")
    return dw, db


def conv3x3_wgrad_f16x3(x, dy, wshape, want_bias=True, alpha=1.0, x_slots=None, dy_slots=None):
    """Backward-weights of a dense 3x3 conv on the fp16 matrix cores, split-fp16 operands (lldwt_conv3x3_wgrad_f16x3_ex).
    x (P,B,cin,h,w), dy (P,B,cout,h,w) -> (dw (P,cout,cin,3,3), dbias (P,cout) or None).  x_slots / dy_slots: the (P,64)
    |max| slots of x / dy (absmax_slots) if the caller has them already -- that pass is then skipped."""
    P, B, cin, h, wd = x.shape
    cout = dy.shape[2]
    if tuple(wshape) != (P, cout, cin, 3, 3) or dy.shape != (P, B, cout, h, wd):
        raise _lib.LLDWTError("conv3x3_wgrad_f16x3: shapes %r %r %r" % (tuple(x.shape), tuple(dy.shape), tuple(wshape)))
    dw = torch.zeros(wshape, device=x.device, dtype=torch.float32)
    db = torch.zeros(P, cout, device=x.device, dtype=torch.float32) if want_bias else None
    slots = torch.empty(P * 128, device=x.device, dtype=torch.float32) if x_slots is None or dy_slots is None else None
    for nm, t in (("x_slots", x_slots), ("dy_slots", dy_slots)):
        if t is not None and (t.numel() != P * 64 or t.dtype != torch.float32):
            raise _lib.LLDWTError("conv3x3_wgrad_f16x3: %s must hold (P, 64) floats" % nm)
    check(_lib.load().lldwt_conv3x3_wgrad_f16x3_ex(_chk(x, "x"), _chk(dy, "dy"), _chk(dw), _opt(db), _opt(slots), _opt(x_slots, "x_slots"),
                                                  _opt(dy_slots, "dy_slots"), cin, cout, P, B, h, wd, float(alpha), _stream()),
          "conv3x3_wgrad_f16x3")
    return dw, db


def act_bwd(dy, y, act):
    dx = torch.empty_like(dy)
    check(_lib.load().lldwt_act_bwd(_chk(dy), _chk(y), _chk(dx), dy.numel(), act, _stream()), "act_bwd")
    return dx


def downsum2(g):
    P, B, Cc, h, w = g.shape
    out = torch.empty(P, B, Cc, h // 2, w // 2, device=g.device, dtype=torch.float32)
    check(_lib.load().lldwt_downsum2(_chk(g), _chk(out), P * B * Cc, h, w, _stream()), "downsum2")
    return out


def gdn(x, beta, gamma, inverse=False, beta_min=1e-6):
    lib = _lib.load()
    P, B, Cc, h, w = x.shape
    y = torch.empty_like(x)
    check(lib.lldwt_gdn(_chk(x, "x"), _chk(y), _chk(beta), _chk(gamma), P, B, Cc, h * w, int(bool(inverse)),
                        float(beta_min), _stream()), "gdn")
    return y


def lower_bound_fwd(x, bound):
    y = torch.empty_like(x)
    check(_lib.load().lldwt_lower_bound_fwd(_chk(x), _chk(y), x.numel(), float(bound), _stream()), "lower_bound_fwd")
    return y


def lower_bound_bwd(x, gy, bound):
    gx = torch.empty_like(x)
    check(_lib.load().lldwt_lower_bound_bwd(_chk(x), _chk(gy), _chk(gx), x.numel(), float(bound), _stream()),
          "lower_bound_bwd")
    return gx


def nonneg_param_fwd(x, minimum):
    y = torch.empty_like(x)
    check(_lib.load().lldwt_nonneg_param_fwd(_chk(x), _chk(y), x.numel(), float(minimum), _stream()), "nonneg_param_fwd")
    return y


def nonneg_param_bwd(x, gy, minimum):
    gx = torch.empty_like(x)
    check(_lib.load().lldwt_nonneg_param_bwd(_chk(x), _chk(gy), _chk(gx), x.numel(), float(minimum), _stream()),
          "nonneg_param_bwd")
    return gx


def quantize(x, noise=None):
    q = torch.empty_like(x)
    check(_lib.load().lldwt_quantize(_chk(x), _opt(noise), _chk(q), x.numel(), _stream()), "quantize")
    return q


def gauss_rate(x, params, noise=None, want_q=False, bit_sum=None):
    """x: (P,B,C,h,w); params: (P,B,2C,h,w) (sigma even, mu odd channels) -> (bits, q or None)."""
    P, B, Cc, h, w = x.shape
    assert params.shape == (P, B, 2 * Cc, h, w)
    bits = torch.empty_like(x)
    q = torch.empty_like(x) if want_q else None
    bs = C.c_void_p(0) if bit_sum is None else C.c_void_p(bit_sum.data_ptr())
    check(_lib.load().lldwt_gauss_rate(_chk(x), _chk(params), _opt(noise), _chk(bits), _opt(q), bs, P * B, Cc, h * w,
                                       _stream()), "gauss_rate")
    return bits, q


def cgp_pack(ws, bs, groups):
    """ws: 4 stacked 1x1 conv weights (P, groups*c_{l+1}, c_l, 1, 1); bs: 4 biases (P, groups*c_{l+1})."""
    lib = _lib.load()
    P = ws[0].shape[0]
    c = [ws[0].shape[2]] + [w.shape[1] // groups for w in ws]
    assert c[4] == 2
    n = lib.lldwt_cgp_packed_floats(c[0], c[1], c[2], c[3], groups)
    packed = torch.empty(P, n, device=ws[0].device, dtype=torch.float32)
    args = []
    for w, b in zip(ws, bs):
        args += [_chk(w, "w"), _chk(b, "b")]
    check(lib.lldwt_cgp_pack(*args, _chk(packed), P, c[0], c[1], c[2], c[3], groups, _stream()), "cgp_pack")
    return packed, tuple(c[:4])


def cgp_rate(cat, x, packed, dims, noise=None, want_params=False, bit_sum=None):
    """cat (P,B,groups*c0,h,w), x (P,B,groups,h,w) -> bits (and (sigma,mu) params if asked)."""
    P, B, G, h, w = x.shape
    assert cat.shape == (P, B, G * dims[0], h, w)
    bits = torch.empty_like(x)
    params = torch.empty(P, B, 2 * G, h, w, device=x.device, dtype=torch.float32) if want_params else None
    bs = C.c_void_p(0) if bit_sum is None else C.c_void_p(bit_sum.data_ptr())
    check(_lib.load().lldwt_cgp_rate(_chk(cat, "cat"), _chk(x, "x"), _opt(noise), _chk(packed, "packed"), _chk(bits),
                                     _opt(params), bs, P, B, h * w, dims[0], dims[1], dims[2], dims[3], G, _stream()),
          "cgp_rate")
    return bits, params


def cgp_rate_ctx(plc, xq, x, packed, dims, K, tap_mask, noise=None, bit_sum=None):
    """Fused cgp stack whose first layer also holds the folded masked context conv (include/lldwt.h lldwt_cgp_rate_ctx).
    plc (P,B,G*cplc,h,w); xq, x (P,B,G,h,w); dims = (cplc + ntaps, c1, c2, c3) as returned by cgp_pack."""
    P, B, G, h, w = x.shape
    ntaps = bin(tap_mask & ((1 << (K * K)) - 1)).count("1")
    cplc = dims[0] - ntaps
    assert plc.shape == (P, B, G * cplc, h, w) and xq.shape == x.shape
    bits = torch.empty_like(x)
    bs = C.c_void_p(0) if bit_sum is None else C.c_void_p(bit_sum.data_ptr())
    check(_lib.load().lldwt_cgp_rate_ctx(_chk(plc, "plc"), _chk(xq, "xq"), _chk(x, "x"), _opt(noise), _chk(packed, "packed"),
                                         _chk(bits), C.c_void_p(0), bs, P, B, h, w, cplc, K, int(tap_mask), dims[1], dims[2],
                                         dims[3], G, _stream()), "cgp_rate_ctx")
    return bits


def cgp_rate_train(cat, x, packed, dims, noise):
    """Training forward of the fused cgp stack: -> (bits, params (P,B,2G,h,w), h1, h2, h3)."""
    P, B, G, h, w = x.shape
    assert cat.shape == (P, B, G * dims[0], h, w)
    bits = torch.empty_like(x)
    params = torch.empty(P, B, 2 * G, h, w, device=x.device, dtype=torch.float32)
    hs = [torch.empty(P, B, G * dims[l], h, w, device=x.device, dtype=torch.float32) for l in (1, 2, 3)]
    check(_lib.load().lldwt_cgp_rate_train(_chk(cat, "cat"), _chk(x, "x"), _opt(noise), _chk(packed, "packed"), _chk(bits),
                                           _chk(params), _chk(hs[0]), _chk(hs[1]), _chk(hs[2]), P, B, h * w, dims[0],
                                           dims[1], dims[2], dims[3], G, _stream()), "cgp_rate_train")
    return bits, params, hs[0], hs[1], hs[2]


def cgp_rate_train_ctx(plc, xq, x, packed, dims, noise, K, tap_mask):
    """Training forward of the fused cgp stack reading its input as the eval path does (lldwt_cgp_rate_train_ctx): plc
    (P,B,G*cplc,h,w) + the live taps of the quantised subband xq (P,B,G,h,w) gathered in the kernel -- no concatenated tensor.
    dims[0] = cplc + number of live taps.  -> (bits, params (P,B,2G,h,w), h1, h2, h3)."""
    P, B, G, h, w = x.shape
    ntaps = bin(int(tap_mask)).count("1")
    cplc = dims[0] - ntaps
    if plc.shape != (P, B, G * cplc, h, w) or xq.shape != (P, B, G, h, w):
        raise _lib.LLDWTError("cgp_rate_train_ctx: shapes %r %r %r" % (tuple(plc.shape), tuple(xq.shape), tuple(x.shape)))
    bits = torch.empty_like(x)
    params = torch.empty(P, B, 2 * G, h, w, device=x.device, dtype=torch.float32)
    hs = [torch.empty(P, B, G * dims[l], h, w, device=x.device, dtype=torch.float32) for l in (1, 2, 3)]
    check(_lib.load().lldwt_cgp_rate_train_ctx(_chk(plc, "plc"), _chk(xq, "xq"), _chk(x, "x"), _opt(noise), _chk(packed, "packed"),
                                               _chk(bits), _chk(params), _chk(hs[0]), _chk(hs[1]), _chk(hs[2]), P, B, h, w, cplc,
                                               K, int(tap_mask), dims[1], dims[2], dims[3], G, _stream()), "cgp_rate_train_ctx")
    return bits, params, hs[0], hs[1], hs[2]


def cgp_bwd_split(dparams, h1, h2, h3, packed_bwd, dims, groups, ntaps):
    """lldwt_cgp_bwd_split -> (dplc (P,B,G*cplc,h,w), dtaps (P,B,G*ntaps,h,w), d1, d2, d3)."""
    P, B, _, h, w = dparams.shape
    cplc = dims[0] - ntaps
    d1, d2, d3 = torch.empty_like(h1), torch.empty_like(h2), torch.empty_like(h3)
    dplc = torch.empty(P, B, groups * cplc, h, w, device=dparams.device, dtype=torch.float32)
    dtaps = torch.empty(P, B, groups * ntaps, h, w, device=dparams.device, dtype=torch.float32)
    check(_lib.load().lldwt_cgp_bwd_split(_chk(dparams), _chk(h1), _chk(h2), _chk(h3), _chk(packed_bwd), _chk(d1), _chk(d2), _chk(d3),
                                          _chk(dplc), _chk(dtaps), P, B, h * w, cplc, ntaps, dims[1], dims[2], dims[3], groups,
                                          _stream()), "cgp_bwd_split")
    return dplc, dtaps, d1, d2, d3


def wgrad1x1_split(xa, xb, dy, groups, want_bias=True):
    """Weight gradient of a grouped 1x1 conv whose input is [xa rows | xb rows] per group (lldwt_wgrad1x1_split):
    xa (P,B,G*ca,h,w), xb (P,B,G*cb,h,w), dy (P,B,cout,h,w) -> (dw (P,cout,ca+cb,1,1), db (P,cout) or None)."""
    P, B, ca_t, h, w = xa.shape
    ca, cb, cout = ca_t // groups, xb.shape[2] // groups, dy.shape[2]
    dw = torch.zeros(P, cout, ca + cb, 1, 1, device=xa.device, dtype=torch.float32)
    db = torch.zeros(P, cout, device=xa.device, dtype=torch.float32) if want_bias else None
    check(_lib.load().lldwt_wgrad1x1_split(_chk(xa, "xa"), _chk(xb, "xb"), _chk(dy, "dy"), _chk(dw), _opt(db), P, B, h * w, ca, cb,
                                           cout, groups, _stream()), "wgrad1x1_split")
    return dw, db


def cgp_pack_bwd(ws, groups):
    """The four forward 1x1 weights (P, groups*c_{l+1}, c_l, 1, 1) -> transposed pack for cgp_bwd."""
    lib = _lib.load()
    P = ws[0].shape[0]
    c = [ws[0].shape[2]] + [w.shape[1] // groups for w in ws]
    n = lib.lldwt_cgp_bwd_packed_floats(c[0], c[1], c[2], c[3], groups)
    packed = torch.empty(P, n, device=ws[0].device, dtype=torch.float32)
    check(lib.lldwt_cgp_pack_bwd(*[_chk(w, "w") for w in ws], _chk(packed), P, c[0], c[1], c[2], c[3], groups, _stream()),
          "cgp_pack_bwd")
    return packed


def cgp_bwd(dparams, h1, h2, h3, packed_bwd, dims, groups):
    """-> (dcat, d1, d2, d3): gradients at the input and at the pre-activation outputs of layers 1..3."""
    P, B, _, h, w = dparams.shape
    d1, d2, d3 = torch.empty_like(h1), torch.empty_like(h2), torch.empty_like(h3)
    dcat = torch.empty(P, B, groups * dims[0], h, w, device=dparams.device, dtype=torch.float32)
    check(_lib.load().lldwt_cgp_bwd(_chk(dparams), _chk(h1), _chk(h2), _chk(h3), _chk(packed_bwd), _chk(d1), _chk(d2),
                                    _chk(d3), _chk(dcat), P, B, h * w, dims[0], dims[1], dims[2], dims[3], groups,
                                    _stream()), "cgp_bwd")
    return dcat, d1, d2, d3


_EB_TABLE = {"key": None, "tab": None, "eb": None}


def factorized_rate(x, eb, noise=None, bit_sum=None):
    """x: (P,B,C,h,w); eb: (P,C,59) packed EntropyBottleneck parameters -> (bits, q)."""
    P, B, Cc, h, w = x.shape
    assert eb.shape == (P, Cc, _lib.EB_FLOATS)
    bits = torch.empty_like(x)
    q = torch.empty_like(x)
    bs = C.c_void_p(0) if bit_sum is None else C.c_void_p(bit_sum.data_ptr())
    if noise is None and not eb.requires_grad:
        # eval: the per-offset bit table depends on the parameters only -- kept while `eb` is the same tensor at the same version
        key = (eb.data_ptr(), eb._version, tuple(eb.shape), eb.device)
        if _EB_TABLE["key"] != key:
            tab = torch.empty(P, Cc, 256, device=eb.device, dtype=torch.float32)
            check(_lib.load().lldwt_factorized_table(_chk(eb), _chk(tab), P, Cc, _stream()), "factorized_table")
            _EB_TABLE["key"], _EB_TABLE["tab"], _EB_TABLE["eb"] = key, tab, eb      # eb kept: its address cannot be reused
        check(_lib.load().lldwt_factorized_rate_tab(_chk(x), _chk(eb), _chk(_EB_TABLE["tab"]), _chk(bits), _chk(q), bs, P, B,
                                                    Cc, h * w, _stream()), "factorized_rate_tab")
        return bits, q
    check(_lib.load().lldwt_factorized_rate(_chk(x), _chk(eb), _opt(noise), _chk(bits), _chk(q), bs, P, B, Cc, h * w,
                                            _stream()), "factorized_rate")
    return bits, q


def sq_err_sum(a, b, out):
    check(_lib.load().lldwt_sq_err_sum(_chk(a), _chk(b), a.numel(), C.c_void_p(out.data_ptr()), _stream()), "sq_err_sum")


def sum_into(x, out):
    check(_lib.load().lldwt_sum(_chk(x), x.numel(), C.c_void_p(out.data_ptr()), _stream()), "sum")


# ------------------------------------------------------------------------------------------------ training support
def gauss_rate_bwd(x, params, noise, gbits):
    P, B, Cc, h, w = x.shape
    dx = torch.empty_like(x)
    dparams = torch.empty_like(params)
    check(_lib.load().lldwt_gauss_rate_bwd(_chk(x), _chk(params), _opt(noise), _chk(gbits), _chk(dx), _chk(dparams), P * B,
                                           Cc, h * w, _stream()), "gauss_rate_bwd")
    return dx, dparams


def factorized_rate_bwd(x, eb, noise, gbits):
    P, B, Cc, h, w = x.shape
    dx = torch.empty_like(x)
    deb = torch.zeros_like(eb)
    check(_lib.load().lldwt_factorized_rate_bwd(_chk(x), _chk(eb), _opt(noise), _chk(gbits), _chk(dx), _chk(deb), P, B, Cc,
                                                h * w, _stream()), "factorized_rate_bwd")
    return dx, deb


def axpby(a, b, alpha, beta=0.0):
    out = torch.empty_like(a)
    check(_lib.load().lldwt_axpby(_chk(a), _opt(b), _chk(out), a.numel(), float(alpha), float(beta), _stream()), "axpby")
    return out


def ycc_to_rgb_bwd(grgb):
    B, _, H, W = grgb.shape
    g = torch.empty(3, B, 1, H, W, device=grgb.device, dtype=torch.float32)
    check(_lib.load().lldwt_ycc_to_rgb_bwd(_chk(grgb), _chk(g), B, H, W, _stream()), "ycc_to_rgb_bwd")
    return g


def lifting_program(Z, H, W, levels, different, block_offset, inverse, Cc, scale=False):
    """-> (list of LiftOp, saved_floats): the step program of the transform (include/lldwt.h lldwt_lifting_program).
    scale=True: with the gain ops of config.scale == 1 (kind 1..4; each keeps its input in the saved buffer)."""
    lib = _lib.load()
    tot = C.c_int64(0)
    sc = int(bool(scale))
    n = lib.lldwt_lifting_program(None, 0, Z, H, W, levels, int(bool(different)), block_offset, int(bool(inverse)), sc, Cc,
                                  C.byref(tot))
    if n < 0:
        check(n, "lifting_program")
    arr = (_lib.LiftOp * n)()
    lib.lldwt_lifting_program(arr, n, Z, H, W, levels, int(bool(different)), block_offset, int(bool(inverse)), sc, Cc,
                              C.byref(tot))
    return list(arr), int(tot.value)


def lifting_forward_train(x, taps, packed, levels, Cc, K, res_weight, linear, different, block_offset, saved, scale_nh=None,
                          scale_nl=None):
    lib = _lib.load()
    P, B, _, H, W = x.shape
    dev = x.device
    ll = torch.empty(P, B, 1, H >> levels, W >> levels, device=dev, dtype=torch.float32)
    yh = [torch.empty(P, B, 3, H >> (i + 1), W >> (i + 1), device=dev, dtype=torch.float32) for i in range(levels)]
    nb = lib.lldwt_lifting_ws_bytes(P * B, H, W, Cc)
    ws = workspace(nb, dev)
    check(lib.lldwt_lifting_forward_train_ex(_chk(x, "x"), _chk(ll), _ptr_array(yh), P, B, H, W, levels, _chk(taps),
                                             _chk(packed), int(packed.shape[1]), int(block_offset), int(bool(different)), Cc, K,
                                             float(res_weight), int(bool(linear)), _opt(scale_nh), _opt(scale_nl),
                                             C.c_void_p(ws.data_ptr()), nb, _chk(saved), _stream()), "lifting_forward_train")
    return ll, yh


def lifting_inverse_train(ll, yh, taps, packed, Cc, K, res_weight, linear, block_offset, saved, scale_nh=None, scale_nl=None):
    lib = _lib.load()
    levels = len(yh)
    P, B, _, hl, wl = ll.shape
    H, W = hl << levels, wl << levels
    x = torch.empty(P, B, 1, H, W, device=ll.device, dtype=torch.float32)
    for t in yh:
        _chk(t, "yh")
    nb = lib.lldwt_lifting_ws_bytes(P * B, H, W, Cc)
    ws = workspace(nb, ll.device)
    check(lib.lldwt_lifting_inverse_train_ex(_chk(ll), _ptr_array(yh), _chk(x), P, B, H, W, levels, _chk(taps), _chk(packed),
                                             int(packed.shape[1]), int(block_offset), Cc, K, float(res_weight),
                                             int(bool(linear)), _opt(scale_nh), _opt(scale_nl), C.c_void_p(ws.data_ptr()), nb,
                                             _chk(saved), _stream()), "lifting_inverse_train")
    return x


def lift_bwd_pre(g_dout, g_din, g, Z, h, w):
    check(_lib.load().lldwt_lift_bwd_pre(g_dout, g_din, _chk(g), Z, h, w, _stream()), "lift_bwd_pre")


def lift_bwd_fin(g, dsk, srcv, g_src, Z, batch, h, w, taps, dtaps, vertical, sign, rw):
    check(_lib.load().lldwt_lift_bwd_fin(_chk(g), _chk(dsk), _chk(srcv), g_src, Z, batch, h, w, _chk(taps), _chk(dtaps),
                                         int(vertical), float(sign), float(rw), _stream()), "lift_bwd_fin")


def lift_step_bwd(g_dst_out, g_dst_in, g_src, saved_step, P, B, h, w, taps, dtaps, packed, packed_plane_stride, dW, Cc, K,
                  rw, sign, vertical, linear, packed_bwd=None, taps_id=None):
    """Whole backward of one lifting step (include/lldwt.h lldwt_lift_step_bwd).  g_*: lldwt_views over the gradient
    buffers; packed: forward pack of this step's block (pointer already offset to the block); dW: the 8 gradient
    tensors (w1,b1,...,w4,b4) of the block, each (P,...), accumulated in place.  packed_bwd (pointer, offset like packed) +
    taps_id ((P,3) of (0,1,0)): the backward-data chain on the fused split-fp16 kernel (lldwt_lift_step_bwd_f16)."""
    lib = _lib.load()
    nb = lib.lldwt_lift_step_bwd_ws_bytes(P * B, h, w, Cc)
    ws = workspace(nb, taps.device)
    if packed_bwd is not None:
        check(lib.lldwt_lift_step_bwd_f16(g_dst_out, g_dst_in, g_src, _chk(saved_step), P, B, h, w, _chk(taps), _chk(dtaps),
                                          packed, packed_plane_stride, *[_chk(t) for t in dW], Cc, K, float(rw), float(sign),
                                          int(bool(vertical)), int(bool(linear)), C.c_void_p(ws.data_ptr()), nb, packed_bwd,
                                          _chk(taps_id), _stream()), "lift_step_bwd_f16")
        return
    check(lib.lldwt_lift_step_bwd(g_dst_out, g_dst_in, g_src, _chk(saved_step), P, B, h, w, _chk(taps), _chk(dtaps),
                                  packed, packed_plane_stride, *[_chk(t) for t in dW], Cc, K, float(rw), float(sign),
                                  int(bool(vertical)), int(bool(linear)), C.c_void_p(ws.data_ptr()), nb, _stream()),
          "lift_step_bwd")


def ew_mul(a, b, scale=1.0):
    out = torch.empty_like(a)
    check(_lib.load().lldwt_ew_mul(_chk(a), _chk(b), _chk(out), a.numel(), float(scale), _stream()), "ew_mul")
    return out


def gdn_apply(x, nrm, inverse):
    y = torch.empty_like(x)
    check(_lib.load().lldwt_gdn_apply(_chk(x), _chk(nrm), _chk(y), x.numel(), int(bool(inverse)), _stream()), "gdn_apply")
    return y


def gdn_apply_bwd(x, nrm, g, inverse):
    dx, dn = torch.empty_like(x), torch.empty_like(x)
    check(_lib.load().lldwt_gdn_apply_bwd(_chk(x), _chk(nrm), _chk(g), _chk(dx), _chk(dn), x.numel(), int(bool(inverse)),
                                          _stream()), "gdn_apply_bwd")
    return dx, dn
