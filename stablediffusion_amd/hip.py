"""ctypes binding of libsdmi_hip.so (the C ABI declared in include/sdmi.h).

The library is built in-tree (``stablediffusion_amd/csrc/libsdmi_hip.so``) by
``__graft_entry__.build()`` / ``make -C stablediffusion_amd/csrc``.  There is no
fallback: if the library is missing, loading raises and every op fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDMI_LIB: development override to A/B a differently built library (tools/ab_build.sh)
LIB_PATH = os.environ.get("SDMI_LIB") or os.path.join(_HERE, "csrc", "libsdmi_hip.so")

SDMI_OK = 0
DTYPE_BF16, DTYPE_F16, DTYPE_F32_PLANES, DTYPE_F16_TO_BF16 = 0, 1, 2, 3


class SdmiError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    """Mirror of ``sdmi_gemm_args`` (include/sdmi.h) - field order and types must match."""
    _fields_ = [
        ("A", C.c_void_p), ("A2", C.c_void_p), ("W", C.c_void_p),
        ("bias", C.c_void_p), ("emb", C.c_void_p), ("res", C.c_void_p),
        ("out", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
        ("lda", C.c_int64), ("lda2", C.c_int64), ("k_split", C.c_int64), ("k_valid", C.c_int64),
        ("ld_res", C.c_int64), ("ld_out", C.c_int64), ("ld_emb", C.c_int64), ("rows_per_batch", C.c_int64),
        ("stride_a", C.c_int64), ("stride_w", C.c_int64), ("stride_out", C.c_int64), ("stride_res", C.c_int64),
        ("mode", C.c_int32),
        ("H", C.c_int32), ("W_in", C.c_int32), ("Cin", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("stride", C.c_int32), ("upsample", C.c_int32),
        ("out_kind", C.c_int32), ("geglu", C.c_int32), ("act", C.c_int32), ("bias_per_row", C.c_int32),
        ("batch", C.c_int32), ("split_k", C.c_int32), ("dtype", C.c_int32),
        ("alpha", C.c_float),
        ("tile_n", C.c_int32),
        ("tile_m", C.c_int32),
        ("asym_pad", C.c_int32),
        ("res_f32", C.c_int32),
        ("stages", C.c_int32),
        ("algo", C.c_int32),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
        ("out_f16", C.c_int32), ("res_f16", C.c_int32), ("a_f16", C.c_int32),
        ("dup_rows", C.c_int32),
        ("ln_colsum", C.c_void_p),
        ("ln_stats", C.c_void_p),
        ("ln_stats_slots", C.c_int32),
        ("row_stats_slots", C.c_int32),
        ("row_stats_out", C.c_void_p),
    ]


ABI_VERSION = 13    # sdmi_abi_version() of the library this binding was written against (include/sdmi.h)

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class XattnArgs(C.Structure):
    """sdmi_xattn_args (include/sdmi.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64), ("M", C.c_int64), ("rows_per_image", C.c_int32),
        ("C", C.c_int32), ("heads", C.c_int32),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
        ("wq", C.c_void_p), ("wo", C.c_void_p), ("bias_o", C.c_void_p),
        ("k", C.c_void_p), ("v", C.c_void_p), ("ldk", C.c_int64), ("ldv", C.c_int64), ("nk", C.c_int32),
        ("scale", C.c_float),
        ("out", C.c_void_p), ("ld_out", C.c_int64),
        ("row_stats_out", C.c_void_p), ("row_stats_slots", C.c_int32),
        ("dtype", C.c_int32),
    ]

# name -> (restype, argtypes); every symbol include/sdmi.h declares
SIGNATURES = {
    "sdmi_status_str": (C.c_char_p, [C.c_int]),
    "sdmi_abi_version": (C.c_int, []),
    "sdmi_diag_gaussian": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp]),
    "sdmi_mul_add": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "sdmi_f32_split3": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _vp, _i32, _vp]),
    "sdmi_f32_groupnorm": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i64, _f32, _i32, _vp]),
    "sdmi_f32_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "sdmi_f32_attention_d64": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f32, _vp]),
    "sdmi_f32_nchw_to_nhwc": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _f32, _vp]),
    "sdmi_f32_attention_vae": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f32, _i32, _vp, _i64, _vp]),
    "sdmi_f32_timestep_embedding": (C.c_int, [_vp, _vp, _i32, _i32, _f32, _vp]),
    "sdmi_f32_embed_tokens": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "sdmi_f32_attention_d64_causal": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f32, _vp]),
    "sdmi_lincomb4": (C.c_int, [_vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _vp, _i64, _vp]),
    "sdmi_gemm_tile_m": (C.c_int32, [C.c_int64, C.c_int64, C.c_int32]),
    "sdmi_gemm": (C.c_int, [C.POINTER(GemmArgs), _vp]),
    "sdmi_gemm_panel_applies": (C.c_int, [C.POINTER(GemmArgs)]),
    "sdmi_gemm_wide_applies": (C.c_int, [C.POINTER(GemmArgs)]),
    "sdmi_gemm_row_stats_slots": (C.c_int32, [C.POINTER(GemmArgs)]),
    "sdmi_row_stats": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "sdmi_xattn_block": (C.c_int, [C.POINTER(XattnArgs), _vp]),
    "sdmi_xattn_block_applies": (C.c_int, [C.POINTER(XattnArgs)]),
    "sdmi_gemm_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "sdmi_pack_tile_n": (_i32, [_i64, _i32]),
    "sdmi_pack_linear_k": (_i64, [_i64, _i32]),
    "sdmi_pack_conv3x3_cin": (_i64, [_i64, _i32]),
    "sdmi_pack_conv3x3_k": (_i64, [_i64, _i32]),
    "sdmi_pack_bytes": (_i64, [_i64, _i64, _i32]),
    "sdmi_pack_linear": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _i32, _i32, _vp]),
    "sdmi_pack_geglu": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i32, _vp]),
    "sdmi_pack_conv3x3": (C.c_int, [_vp, _i64, _i64, _vp, _i32, _i32, _vp]),
    "sdmi_pack_linear_host": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _i32, _i32]),
    "sdmi_pack_geglu_host": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i32]),
    "sdmi_pack_conv3x3_host": (C.c_int, [_vp, _i64, _i64, _vp, _i32, _i32]),
    "sdmi_groupnorm": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i64, _f32, _i32, _i32, _vp, _i64, _vp]),
    "sdmi_groupnorm_workspace_bytes": (_i64, [_i32, _i64]),
    "sdmi_groupnorm_noalloc": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i64, _f32, _i32, _i32, _vp]),
    "sdmi_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _i32, _vp]),
    "sdmi_attention_d64": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f32, _i32, _vp]),
    "sdmi_attention_vae": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f32, _i32, _i32, _vp, _i64, _vp]),
    "sdmi_attention_vae_workspace_bytes": (_i64, [_i32, _i32]),
    "sdmi_softmax_rows": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "sdmi_softmax_cols": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _i32, _vp]),
    "sdmi_softmax_cols_scratch_bytes": (_i64, [_i32, _i64]),
    "sdmi_nchw_f32_to_nhwc": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _f32, _i32, _vp]),
    "sdmi_cast_f32": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "sdmi_timestep_embedding": (C.c_int, [_vp, _vp, _i32, _i32, _f32, _i32, _vp]),
    "sdmi_ddim_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _f32, _f32, _f32, _i32, _vp]),
    "sdmi_image_to_u8": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _vp]),
    "sdmi_probe_mfma": (C.c_int, [_vp, _i32, C.POINTER(C.c_double), _vp]),
    "sdmi_probe_copy": (C.c_int, [_vp, _vp, _i64, _vp]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises SdmiError when the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SdmiError(
                f"{LIB_PATH} not found: build it with `make -C stablediffusion_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU/PyTorch fallback for the HIP path")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.sdmi_abi_version() != ABI_VERSION:
            raise SdmiError(f"{LIB_PATH} has ABI version {handle.sdmi_abi_version()}, this package expects "
                            f"{ABI_VERSION} (GemmArgs layout, entry points): rebuild with `make -C stablediffusion_amd/csrc`")
        _lib = handle
    return _lib


def check(status, what=""):
    if status != SDMI_OK:
        msg = lib().sdmi_status_str(status).decode()
        raise SdmiError(f"{what or 'sdmi call'} failed: {msg} ({status})")
