"""ctypes binding of libicaf.so (C ABI declared in include/icaf.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  `lib()` raises if the shared object is
missing or does not export every symbol of the header, so a GPU box that silently lost the extension fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICAF_LIB") or os.path.join(_HERE, "lib", "libicaf.so")   # ICAF_LIB: A/B a variant build (tools/)

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2


class ConvArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("res", C.c_void_p),
        ("x_gs", C.c_longlong), ("w_gs", C.c_longlong), ("bias_gs", C.c_longlong), ("y_gs", C.c_longlong),
        ("res_gs", C.c_longlong),
        ("groups", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("ldx", C.c_int),
        ("Ho", C.c_int), ("Wo", C.c_int), ("Cout", C.c_int), ("ldy", C.c_int),
        ("kh", C.c_int), ("kw", C.c_int), ("sh", C.c_int), ("sw", C.c_int), ("ph", C.c_int), ("pw", C.c_int),
        ("ldr", C.c_int), ("Kp", C.c_int), ("act", C.c_int), ("dtype", C.c_int), ("out_dtype", C.c_int),
        ("alpha_acc", C.c_float * 2), ("alpha_res", C.c_float * 2),
        ("tile", C.c_int),
        ("pre", C.c_void_p), ("pre_h", C.c_int), ("pre_w", C.c_int), ("ldpre", C.c_int), ("pre_mode", C.c_int),
        ("w2", C.c_void_p), ("bias2", C.c_void_p), ("y2", C.c_void_p),
        ("w2_gs", C.c_longlong), ("bias2_gs", C.c_longlong), ("y2_gs", C.c_longlong),
        ("Kp2", C.c_int), ("Cout2", C.c_int), ("ldy2", C.c_int), ("chain_keep", C.c_int),
        ("wf", C.c_void_p), ("wf_gs", C.c_longlong),
        ("x2", C.c_void_p), ("x2_gs", C.c_longlong), ("ldx2", C.c_int), ("reserved2", C.c_int),
    ]


class BneckArgs(C.Structure):
    _fields_ = [("conv", ConvArgs), ("w1", C.c_void_p), ("bias1", C.c_void_p), ("w1_gs", C.c_longlong),
                ("bias1_gs", C.c_longlong), ("Kp1", C.c_int), ("shape", C.c_int),
                ("x2", C.c_void_p), ("x2_gs", C.c_longlong), ("ldx2", C.c_int), ("reserved", C.c_int)]


class Stem2Args(C.Structure):
    _fields_ = [("img", C.c_void_p), ("img_u8", C.c_int), ("ctot", C.c_int),
                ("dtype", C.c_int), ("nstreams", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("w0", C.c_void_p), ("bias0", C.c_void_p), ("w0_gs", C.c_longlong), ("bias0_gs", C.c_longlong),
                ("Kp0", C.c_int), ("C0", C.c_int),
                ("w1", C.c_void_p), ("bias1", C.c_void_p), ("w1_gs", C.c_longlong), ("bias1_gs", C.c_longlong),
                ("Kp1", C.c_int), ("C1", C.c_int),
                ("w2", C.c_void_p), ("bias2", C.c_void_p), ("w2_gs", C.c_longlong), ("bias2_gs", C.c_longlong),
                ("Kp2", C.c_int), ("C2", C.c_int),
                ("y", C.c_void_p), ("y_gs", C.c_longlong), ("ldy", C.c_int), ("reserved", C.c_int)]


class DmffArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("qkv", C.c_void_p), ("y", C.c_void_p),
                ("wqkv", C.c_void_p), ("bqkv", C.c_void_p), ("wo", C.c_void_p), ("bo", C.c_void_p),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("ln_attn_gamma", C.c_void_p * 2), ("ln_attn_beta", C.c_void_p * 2),
                ("ln_mlp_gamma", C.c_void_p), ("ln_mlp_beta", C.c_void_p),
                ("wqkv_gs", C.c_longlong), ("bqkv_gs", C.c_longlong), ("wo_gs", C.c_longlong), ("bo_gs", C.c_longlong),
                ("w1_gs", C.c_longlong), ("b1_gs", C.c_longlong), ("w2_gs", C.c_longlong), ("b2_gs", C.c_longlong),
                ("x_gs", C.c_longlong), ("y_gs", C.c_longlong),
                ("dtype", C.c_int), ("B", C.c_int), ("N", C.c_int), ("C", C.c_int), ("heads", C.c_int), ("Kp", C.c_int),
                ("Kp4", C.c_int), ("hidden", C.c_int), ("ldy", C.c_int), ("reserved", C.c_int),
                ("eps_attn", C.c_float), ("eps_mlp", C.c_float),
                ("coef_res_attn", C.c_float * 2), ("coef_acc_attn", C.c_float * 2),
                ("coef_res_mlp", C.c_float * 2), ("coef_acc_mlp", C.c_float * 2), ("debug_clock", C.c_void_p),
                ("x32", C.c_void_p), ("y32", C.c_void_p)]          # fp32 residual stream across iterations (icaf.h)


_p, _i, _ll, _f, _sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t
# symbol -> (restype, argtypes); must list every function declared in include/icaf.h
SIGNATURES = {
    "icaf_last_error": (C.c_char_p, []),
    "icaf_set_option": (_i, [C.c_char_p, _i]),
    "icaf_version": (_i, []),
    "icaf_device_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "icaf_preprocess_nchw": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "icaf_preprocess_u8": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "icaf_stem": (_i, [_p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _ll, _ll, _ll, _p]),
    "icaf_stem2": (_i, [C.POINTER(Stem2Args), _p]),
    "icaf_conv2d": (_i, [C.POINTER(ConvArgs), _p]),
    "icaf_bottleneck": (_i, [C.POINTER(BneckArgs), _p]),
    "icaf_conv2d_kernel_name": (_i, [C.POINTER(ConvArgs), C.c_char_p, _i]),
    "icaf_sppf_pool": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "icaf_upsample_nearest": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "icaf_copy_channels": (_i, [_p, _i, _p, _i, _i, _ll, _i, _p]),
    "icaf_axpby": (_i, [_p, _i, _p, _i, _p, _i, _i, _ll, _i, _f, _f, _p]),
    "icaf_dmff_pool_tokens": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                   _f, _f, _f, _f, _p]),
    "icaf_layernorm": (_i, [_p, _p, _p, _p, _p, _p, _i, _ll, _i, _i, _f, _p]),
    "icaf_cross_attention": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "icaf_dmff_ln_qkv": (_i, [C.POINTER(DmffArgs), _p]),
    "icaf_dmff_attn_mlp": (_i, [C.POINTER(DmffArgs), _p]),
    "icaf_dmff_wide_ln_qkv": (_i, [C.POINTER(DmffArgs), _p]),
    "icaf_dmff_wide_proj_mlp": (_i, [C.POINTER(DmffArgs), _p, _p]),
    "icaf_dmff_wide_proj_mlp_split": (_i, [C.POINTER(DmffArgs), _p, _p, _i, _p]),
    "icaf_dmff_wide_reduce": (_i, [C.POINTER(DmffArgs), _p, _i, _p]),
    "icaf_dmff_attn_mlp_lds_bytes": (_i, [_i, _i, _i, _i, C.POINTER(_sz)]),
    "icaf_dmff_upsample_merge": (_i, [_p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "icaf_detect_decode": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _ll, _ll, _f, C.POINTER(_f), _p]),
    "icaf_detect_conv": (_i, [C.POINTER(ConvArgs), _p, _p, _p, _i, _i, _ll, _ll, _f, C.POINTER(_f), _p]),
    "icaf_match_predictions": (_i, [_p, _p, _i, _i, _p, _p, _i, _p, _p, _i, _p, _p, _p]),
    "icaf_nms_workspace_bytes": (_i, [_i, _ll, _i, _i, C.POINTER(_sz)]),
    "icaf_nms": (_i, [_p, _i, _ll, _i, _f, _f, _i, _i, C.POINTER(_i), _i, _i, _i, _f, _p, _p, _p, _p, _sz, _p]),
    "icaf_graph_begin": (_i, [_p]),
    "icaf_graph_end": (_i, [_p, C.POINTER(_p)]),
    "icaf_graph_launch": (_i, [_p, _p]),
    "icaf_graph_destroy": (_i, [_p]),
    "icaf_event_create": (_i, [C.POINTER(_p)]),
    "icaf_event_record": (_i, [_p, _p]),
    "icaf_stream_wait_event": (_i, [_p, _p]),
    "icaf_event_elapsed_ms": (_i, [_p, _p, C.POINTER(_f)]),
    "icaf_event_destroy": (_i, [_p]),
    "icaf_stream_sync": (_i, [_p]),
}

_LIB = None


class IcafError(RuntimeError):
    pass


def lib():
    """Load libicaf.so once; raise (never fall back) when it is absent or incomplete."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise IcafError(f"{LIB_PATH} not found — build it with `python -m icafusion_amd.build` "
                            "(hipcc --offload-arch=gfx950); icafusion_amd has no CPU / PyTorch fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise IcafError(f"libicaf.so does not export {name}; rebuild it") from e
            fn.restype, fn.argtypes = res, args
        from .options import OPT                       # the library's probe knobs come from the one options object, not from its environment
        for name, value in OPT.lib_options().items():
            if handle.icaf_set_option(name.encode(), int(value)) != 0:
                raise IcafError(f"icaf_set_option({name}): {handle.icaf_last_error().decode()}")
        _LIB = handle
    return _LIB


def check(status, what=""):
    if status != 0:
        msg = lib().icaf_last_error()
        raise IcafError(f"{what} failed ({status}): {msg.decode() if msg else '?'}")
