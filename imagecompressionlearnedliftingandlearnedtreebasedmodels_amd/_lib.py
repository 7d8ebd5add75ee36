"""ctypes binding of the C-ABI library ``liblldwt.so`` (include/lldwt.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails this module raises.
PyTorch is plumbing only (device memory, streams); signatures carry raw pointers and sizes.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblldwt.so")


class LLDWTError(RuntimeError):
    pass


class View(C.Structure):
    """lldwt_view: strided (Z,h,w) single-channel view (include/lldwt.h)."""
    _fields_ = [("p", C.c_void_p), ("sz", C.c_int64), ("sy", C.c_int64), ("sx", C.c_int64)]


class LiftOp(C.Structure):
    """lldwt_lift_op (include/lldwt.h)."""
    _fields_ = [("kind", C.c_int32), ("buf_src", C.c_int32), ("buf_din", C.c_int32), ("buf_dout", C.c_int32),
                ("off_src", C.c_int64), ("sz_src", C.c_int64), ("sy_src", C.c_int64), ("sx_src", C.c_int64),
                ("off_din", C.c_int64), ("sz_din", C.c_int64), ("sy_din", C.c_int64), ("sx_din", C.c_int64),
                ("off_dout", C.c_int64), ("sz_dout", C.c_int64), ("sy_dout", C.c_int64), ("sx_dout", C.c_int64),
                ("h", C.c_int32), ("w", C.c_int32), ("vertical", C.c_int32), ("tap", C.c_int32), ("block", C.c_int32),
                ("is_u", C.c_int32), ("sign", C.c_float), ("pad_", C.c_int32), ("saved_off", C.c_int64)]


class ConvDesc(C.Structure):
    """lldwt_conv_desc (include/lldwt.h)."""
    _fields_ = [("cin", C.c_int), ("cout", C.c_int), ("K", C.c_int), ("groups", C.c_int), ("act", C.c_int),
                ("upsample2", C.c_int), ("transposed", C.c_int), ("tap_mask", C.c_uint32), ("oc_block", C.c_int),
                ("oc_stride", C.c_int), ("oc_off", C.c_int), ("ytot", C.c_int), ("ic_block", C.c_int),
                ("ic_stride", C.c_int), ("ic_off", C.c_int), ("xtot", C.c_int), ("epi", C.c_int)]


ACT_NONE, ACT_TANH, ACT_LRELU, ACT_RELU = 0, 1, 2, 3
EPI_NONE, EPI_TANH_BWD, EPI_LRELU_BWD = 0, 1, 2
EB_FLOATS = 59

_p, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol declared in include/lldwt.h
SIGNATURES = {
    "lldwt_last_error": (C.c_char_p, []),
    "lldwt_version": (_i, []),
    "lldwt_device_ok": (_i, []),
    "lldwt_rgb_to_ycc": (_i, [_p, _p, _i64, _i64, _i64, _p]),
    "lldwt_u8hwc_to_f32chw": (_i, [_p, _p, _i64, _i64, _i64, _p]),
    "lldwt_ycc_to_rgb": (_i, [_p, _p, _i64, _i64, _i64, _i, _p]),
    "lldwt_pblock_packed_floats": (_i64, [_i, _i]),
    "lldwt_set_lift_mode": (_i, [_i]),
    "lldwt_set_diagnostics": (_i, [_i, _p, _i64, _i]),
    "lldwt_wgrad16_f16x3": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _f, _i, _p]),
    "lldwt_cgp16_wavefront_step": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _i, C.c_uint32, _i, _i64, _i64, _p]),
    "lldwt_wavefront_apply": (_i, [_p, _p, _p, _i64, _i64, _i64, _i64, _i, _i, _i, _i64, _i64, _p]),
    "lldwt_rans_decode_multi": (_i, [C.POINTER(_p), _i64, _p, _i64, _i64, _p, C.c_int32, C.c_int32, _p, _p, _p]),
    "lldwt_train_lift_f16": (_i, []),
    "lldwt_set_precision": (_i, [_i]),
    "lldwt_get_precision": (_i, []),
    "lldwt_set_cdf97_short_levels": (_i, [_i]),
    "lldwt_get_lift_mode": (_i, []),
    "lldwt_pack_pblock": (_i, [_p] * 9 + [_i, _i, _i, _p]),
    "lldwt_pack_pblock_train": (_i, [_p] * 9 + [_i, _i, _i, _p]),
    "lldwt_pack_pblock_seq": (_i, [_p] * 9 + [_i, _i, _i, _p]),
    "lldwt_lift_step_ws_bytes": (_i64, [_i64, _i64, _i64, _i]),
    "lldwt_lift_step": (_i, [View, View, View, _i64, _i64, _i64, _i64, _p, _p, _i, _i, _i, _f, _f, _i, _p, _i64, _p]),
    "lldwt_lifting_ws_bytes": (_i64, [_i64, _i64, _i64, _i]),
    "lldwt_lifting_forward": (_i, [_p, _p, C.POINTER(_p), _i64, _i64, _i64, _i64, _i, _p, _p, _i, _i, _i, _i, _i, _f, _i,
                                   _p, _p, _p, _i64, _p]),
    "lldwt_lifting_inverse": (_i, [_p, C.POINTER(_p), _p, _i64, _i64, _i64, _i64, _i, _p, _p, _i, _i, _i, _i, _f, _i,
                                   _p, _p, _p, _i64, _p]),
    "lldwt_lifting_program": (_i, [C.POINTER(LiftOp), _i, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i, C.POINTER(_i64)]),
    "lldwt_lifting_forward_train": (_i, [_p, _p, C.POINTER(_p), _i64, _i64, _i64, _i64, _i, _p, _p, _i, _i, _i, _i, _i, _f,
                                         _i, _p, _i64, _p, _p]),
    "lldwt_lifting_inverse_train": (_i, [_p, C.POINTER(_p), _p, _i64, _i64, _i64, _i64, _i, _p, _p, _i, _i, _i, _i, _f, _i,
                                         _p, _i64, _p, _p]),
    "lldwt_lifting_forward_train_ex": (_i, [_p, _p, C.POINTER(_p), _i64, _i64, _i64, _i64, _i, _p, _p, _i, _i, _i, _i, _i, _f,
                                            _i, _p, _p, _p, _i64, _p, _p]),
    "lldwt_lifting_inverse_train_ex": (_i, [_p, C.POINTER(_p), _p, _i64, _i64, _i64, _i64, _i, _p, _p, _i, _i, _i, _i, _f, _i,
                                            _p, _p, _p, _i64, _p, _p]),
    "lldwt_lift_bwd_pre": (_i, [View, View, _p, _i64, _i64, _i64, _p]),
    "lldwt_lift_bwd_fin": (_i, [_p, _p, _p, View, _i64, _i64, _i64, _i64, _p, _p, _i, _f, _f, _p]),
    "lldwt_lift_step_bwd_ws_bytes": (_i64, [_i64, _i64, _i64, _i]),
    "lldwt_lift_step_bwd": (_i, [View, View, View, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _i64] + [_p] * 8 +
                            [_i, _i, _f, _f, _i, _i, _p, _i64, _p]),
    "lldwt_lift_step_bwd_f16": (_i, [View, View, View, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _i64] + [_p] * 8 +
                                [_i, _i, _f, _f, _i, _i, _p, _i64, _p, _p, _p]),
    "lldwt_bwd_lift_f16": (_i, []),
    "lldwt_pack_pblock_bwd_ws_bytes": (_i64, [_i]),
    "lldwt_pack_pblock_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _p]),
    "lldwt_subband_mlp_bwd_w_ws_bytes": (_i64, [_i64, _i, _i64]),
    "lldwt_subband_mlp_bwd_w": (_i, [_p, _p, _p, _i64, _i64, _i, _i64, _i] + [_p] * 7 + [_p] * 8 + [_p, _i64, _p]),
    "lldwt_subband_mlp_bwd": (_i, [_p] * 9 + [_i64, _i64, _i, _i64, _i] + [_p] * 7 + [_p]),
    "lldwt_subband_mlp": (_i, [_p, _p, _i64, _i64, _i, _i64, _i] + [_p] * 8 + [_i, _p]),
    "lldwt_conv_packed_floats": (_i64, [C.POINTER(ConvDesc)]),
    "lldwt_conv_pack": (_i, [_p, _p, C.POINTER(ConvDesc), _i64, _p]),
    "lldwt_conv_pack_ex": (_i, [_p, _p, C.POINTER(ConvDesc), _i64, _i, _p]),
    "lldwt_conv2d": (_i, [_p, _p, _p, _p, _p, _p, C.POINTER(ConvDesc), _i64, _i64, _i64, _i64, _p]),
    "lldwt_conv2d_absmax": (_i, [_p, _p, _p, _p, _p, _p, _p, C.POINTER(ConvDesc), _i64, _i64, _i64, _i64, _p]),
    "lldwt_conv2d_f16out": (_i, [_p, _p, _p, _p, _p, C.POINTER(ConvDesc), _i64, _i64, _i64, _i64, _p]),
    "lldwt_conv3x3_f16in": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i64, _i64, _i64, _i64, _p]),
    "lldwt_conv2d_wgrad": (_i, [_p, _p, _p, _p, C.POINTER(ConvDesc), _i64, _i64, _i64, _i64, _p]),
    "lldwt_conv2d_wgrad_ex": (_i, [_p, _p, _p, _p, C.POINTER(ConvDesc), _i64, _i64, _i64, _i64, _f, _i, _p]),
    "lldwt_conv3x3_wgrad_f16x3": (_i, [_p, _p, _p, _p, _p, _i, _i, _i64, _i64, _i64, _i64, _f, _p]),
    "lldwt_conv3x3_wgrad_f16x3_ex": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i64, _i64, _i64, _f, _p]),
    "lldwt_conv_f16x3_packed_bytes": (_i64, [_i, _i]),
    "lldwt_plc_shape16": (_i, []),
    "lldwt_plc_fused_pack1_bytes": (_i64, [_i]),
    "lldwt_plc_fused_pack1": (_i, [_p, _p, _p, _i, _i64, _p]),
    "lldwt_plc_fused": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i64, _i64, _i64, _i64, _p]),
    "lldwt_conv_f16x3_pack": (_i, [_p, _p, _i, _i, _i64, _p]),
    "lldwt_absmax_slots": (_i, [_p, _i64, _i64, _p, _p]),
    "lldwt_conv3x3_f16x3": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i64, _i64, _i64, _i64, _p]),
    "lldwt_act_bwd": (_i, [_p, _p, _p, _i64, _i, _p]),
    "lldwt_downsum2": (_i, [_p, _p, _i64, _i64, _i64, _p]),
    "lldwt_conv2d_direct": (_i, [_p, _p, _p, _p, C.POINTER(ConvDesc), _i64, _i64, _i64, _i64, _p]),
    "lldwt_gdn": (_i, [_p, _p, _p, _p, _i64, _i64, _i, _i64, _i, _f, _p]),
    "lldwt_ew_mul": (_i, [_p, _p, _p, _i64, _f, _p]),
    "lldwt_gdn_apply": (_i, [_p, _p, _p, _i64, _i, _p]),
    "lldwt_gdn_apply_bwd": (_i, [_p, _p, _p, _p, _p, _i64, _i, _p]),
    "lldwt_lower_bound_fwd": (_i, [_p, _p, _i64, _f, _p]),
    "lldwt_lower_bound_bwd": (_i, [_p, _p, _p, _i64, _f, _p]),
    "lldwt_nonneg_param_fwd": (_i, [_p, _p, _i64, _f, _p]),
    "lldwt_nonneg_param_bwd": (_i, [_p, _p, _p, _i64, _f, _p]),
    "lldwt_gauss_rate": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _i64, _p]),
    "lldwt_cgp_packed_floats": (_i64, [_i, _i, _i, _i, _i]),
    "lldwt_cgp_pack": (_i, [_p] * 9 + [_i64, _i, _i, _i, _i, _i, _p]),
    "lldwt_cgp_rate": (_i, [_p] * 7 + [_i64, _i64, _i64, _i, _i, _i, _i, _i, _p]),
    "lldwt_cgp_rate_ctx": (_i, [_p] * 8 + [_i64, _i64, _i64, _i64, _i, _i, C.c_uint32, _i, _i, _i, _i, _p]),
    "lldwt_cgp16_packed_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "lldwt_cgp16_pack": (_i, [_p] * 9 + [_i64, _i, _i, _i, _i, _i, _p]),
    "lldwt_cgp16_params": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _i, C.c_uint32, _p]),
    "lldwt_cgp16_params_train": (_i, [_p] * 7 + [_i64, _i64, _i64, _i64, _i, _i, C.c_uint32, _p]),
    "lldwt_cgp16_bwd_packed_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "lldwt_cgp16_pack_bwd": (_i, [_p] * 5 + [_i64, _i, _i, _i, _i, _i, _p]),
    "lldwt_cgp16_bwd": (_i, [_p] * 10 + [_i64, _i64, _i64, _i, _p]),
    "lldwt_cgp_rate_train": (_i, [_p] * 9 + [_i64, _i64, _i64, _i, _i, _i, _i, _i, _p]),
    "lldwt_cgp_bwd_packed_floats": (_i64, [_i, _i, _i, _i, _i]),
    "lldwt_cgp_pack_bwd": (_i, [_p] * 5 + [_i64, _i, _i, _i, _i, _i, _p]),
    "lldwt_cgp_bwd": (_i, [_p] * 9 + [_i64, _i64, _i64, _i, _i, _i, _i, _i, _p]),
    "lldwt_cgp_rate_train_ctx": (_i, [_p] * 10 + [_i64, _i64, _i64, _i64, _i, _i, C.c_uint32, _i, _i, _i, _i, _p]),
    "lldwt_cgp_bwd_split": (_i, [_p] * 10 + [_i64, _i64, _i64, _i, _i, _i, _i, _i, _i, _p]),
    "lldwt_wgrad1x1_split": (_i, [_p] * 5 + [_i64, _i64, _i64, _i, _i, _i, _i, _p]),
    "lldwt_gauss_rate_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _i64, _p]),
    "lldwt_axpby": (_i, [_p, _p, _p, _i64, _f, _f, _p]),
    "lldwt_ycc_to_rgb_bwd": (_i, [_p, _p, _i64, _i64, _i64, _p]),
    "lldwt_quantize": (_i, [_p, _p, _p, _i64, _p]),
    "lldwt_factorized_rate": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i, _i64, _p]),
    "lldwt_factorized_table": (_i, [_p, _p, _i64, _i, _p]),
    "lldwt_factorized_rate_tab": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i, _i64, _p]),
    "lldwt_factorized_rate_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i, _i64, _p]),
    "lldwt_pmf_to_quantized_cdf": (_i, [_p, _i, _i, _p]),
    "lldwt_rans_encode": (_i64, [_p, _p, _i64, _p, C.c_int32, C.c_int32, _p, _p, _p, _i64]),
    "lldwt_rans_decoder_new": (_p, [_p, _i64]),
    "lldwt_rans_decode": (_i, [_p, _p, _i64, _p, C.c_int32, C.c_int32, _p, _p, _p]),
    "lldwt_rans_decoder_free": (None, [_p]),
    "lldwt_sq_err_sum": (_i, [_p, _p, _i64, _p, _p]),
    "lldwt_sum": (_i, [_p, _i64, _p, _p]),
    "lldwt_cdf97_ws_bytes": (_i64, [_i64, _i64, _i64]),
    "lldwt_cdf97_forward": (_i, [_p, _p, C.POINTER(_p), _i64, _i64, _i64, _i, _p, _i64, _p]),
    "lldwt_cdf97_inverse": (_i, [_p, C.POINTER(_p), _p, _i64, _i64, _i64, _i, _p, _i64, _p]),
    "lldwt_cdf97_forward_ex": (_i, [_p, _p, C.POINTER(_p), _i64, _i64, _i64, _i, _i, _p, _i64, _p]),
    "lldwt_cdf97_inverse_ex": (_i, [_p, C.POINTER(_p), _p, _i64, _i64, _i64, _i, _i, _p, _i64, _p]),
}

_lib = None


def load():
    """Load liblldwt.so (built by csrc/build.sh / __graft_entry__.build()); raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LLDWTError(
            "HIP library %s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    mode = os.environ.get("LLDWT_LIFT_MODE", "f16x3")
    if mode not in ("f16x3", "f32"):
        raise LLDWTError("LLDWT_LIFT_MODE must be 'f16x3' or 'f32' (got %r)" % mode)
    lib.lldwt_set_lift_mode(1 if mode == "f16x3" else 0)
    prec = os.environ.get("LLDWT_PRECISION", "f16x3")
    if prec not in PRECISIONS:
        raise LLDWTError("LLDWT_PRECISION must be one of %s (got %r)" % (sorted(PRECISIONS), prec))
    lib.lldwt_set_precision(PRECISIONS[prec])
    return lib


# arithmetic of the eval path's matrix kernels (lldwt_set_precision): three fp16 MFMA products per fp32 MAC (default, fp32-level
# accuracy), or ONE product on fp16 / bf16 operands (BASELINE configs[4] / configs[1]; tolerance class 1e-2)
PRECISIONS = {"f16x3": 0, "fp16": 1, "bf16": 2}


def check(rc, what):
    if rc != 0:
        msg = load().lldwt_last_error()
        raise LLDWTError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
