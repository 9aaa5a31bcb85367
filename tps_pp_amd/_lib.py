"""ctypes binding of libtpspp_hip.so (C ABI: include/tpspp.h).

The product has no fallback: if the library is missing or fails to load, importing an op raises.
PyTorch is used only for device memory and streams -- tensors cross this boundary as raw pointers.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtpspp_hip.so")
ABI_VERSION = 5

_f = ctypes.c_void_p       # device pointers travel as integers
_i = ctypes.c_int

_SIGNATURES = {
    "tpspp_abi_version": ([], _i),
    "tpspp_last_error": ([], ctypes.c_char_p),
    "tpspp_solve_T": ([_f, _f, _i, _i, _f, _f], _i),
    "tpspp_build_grid": ([_f, _i, _f, _f, _f, _i, _i, _i, _f, _f], _i),
    "tpspp_grid_sample": ([_f, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f], _i),
    "tpspp_transpose_p_hat": ([_f, _i, _i, _i, _f, _f], _i),
    "tpspp_table_mirror_symmetry": ([_f, _i, _i, _i, _i], _i),
    "tpspp_prepared_table_floats": ([_i, _i, _i], ctypes.c_size_t),
    "tpspp_prepare_mirror_table": ([_f, _i, _i, _i, _i, _f, _f], _i),
    "tpspp_warp_fwd": ([_f, _i, _i, _i, _f, _i, _i, _i, _f, _f, _f, _f, _i, _f, _f, _i, _i, _i, _i, _i,
                        _f, _f, _f, _f, _f], _i),
    "tpspp_warp_plan_create": ([_f, _i, _i, _i, _f, _i, _i, _i, _f, _f, _f, _f, _i, _f, _f, _i, _i, _i, _i, _i,
                                _f, _f, _f, _f, _f, _f], _i),
    "tpspp_warp_plan_run": ([_f], _i),
    "tpspp_warp_plan_run_on": ([_f, _f], _i),
    "tpspp_warp_plan_destroy": ([_f], None),
    "tpspp_conv2d_fwd": ([_f, _f, _i, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _f], _i),
    "tpspp_conv2d_bf16_fwd": ([_f, _f, _i, _f, _f, _f, _i, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _f], _i),
    "tpspp_conv_bf16_chunk_channels": ([_i], _i),
    "tpspp_dgab_fwd": ([_f] * 16 + [_i, _i, _f], _i),
    "tpspp_dgab_bf16_fwd": ([_f] * 16 + [_i, _i, _i, _f], _i),
    "tpspp_score_fwd": ([_f, _f, _f, _f, _f, _f, ctypes.c_float, _f, _i, _i, _f], _i),
    "tpspp_score_x3_fwd": ([_f, _f, _f, _f, _f, _f, ctypes.c_float, _f, _i, _i, _f], _i),
    "tpspp_front_fwd": ([_f] * 15 + [_i, _i, _i, _f], _i),
    "tpspp_front_bf16_fwd": ([_f] * 15 + [_i, _i, _i, _i, _i, _f], _i),
    "tpspp_down_fused_bf16_fwd": ([_f] * 6 + [_i, _i, _i, _i, _f], _i),
    "tpspp_down_fused_x3_fwd": ([_f] * 6 + [_i, _i, _i, _i, _f], _i),
    "tpspp_down_fused_f32_fwd": ([_f] * 6 + [_i, _i, _i, _i, _f], _i),
    "tpspp_token_gemm_bf16_fwd": ([_f] * 5 + [_i, _i, _i, _i, _i, _i, _f], _i),
    "tpspp_cbam_fwd": ([_f, _f, _f, _f, _f, _f, _i, _f], _i),
    "tpspp_tpe_points_fwd": ([_f] * 13 + [_i, _f], _i),
    "tpspp_maxpool2x2_fwd": ([_f, _i, _i, _i, _i, _f, _f], _i),
    "tpspp_global_avgpool_fwd": ([_f, _i, _i, _i, _i, _f, _f], _i),
    "tpspp_conv_chunk_channels": ([_i], _i),
    "tpspp_conv_set_tuning": ([_i], _i),
    "tpspp_warp_set_tuning": ([_i, _i, _i, _i], _i),
    "tpspp_warp_set_trace": ([_f], _i),
    "tpspp_head_set_trace": ([_f], _i),
    "tpspp_lab_occupy": ([_i, _i, _i, _f], _i),
    "tpspp_warp_bwd_workspace_floats": ([_i, _i, _i], ctypes.c_size_t),
    "tpspp_warp_bwd_set_accumulator": ([_i], _i),
    "tpspp_warp_bwd": ([_f, _f, _i, _i, _i, _f, _f, _i, _i, _i, _f, _f, _f, _f, _i, _f, _f, _f, _i, _i, _i, _i, _i,
                        _f, _f, _f, _f, _f, ctypes.c_size_t, _f], _i),
    "tpspp_transpose2d": ([_f, _i, _i, _f, _f], _i),
    "tpspp_layernorm_cm_fwd": ([_f, _f, _f, _i, _i, ctypes.c_float, _f, _f], _i),
    "tpspp_attn_enc_fwd": ([_f, _i, _i, _i, _f, _f, _f], _i),
    "tpspp_linear_ln_fwd": ([_f, _i, _i, ctypes.c_float, _f, _f, _i, _f, _i, _f, _i, _f, _f], _i),
    "tpspp_resize_normalize_fwd": ([_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _i, _f], _i),
    "tpspp_nrtr_encoder_workspace": ([_i, _i, _i, _i], ctypes.c_size_t),
    "tpspp_nrtr_decoder_workspace": ([_i] * 7, ctypes.c_size_t),
    "tpspp_nrtr_encoder_fwd": ([_f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, ctypes.c_size_t, _f, _f, _i, _f], _i),
    "tpspp_nrtr_decoder_fwd": ([_f, _i, _i, _i, _i, _i, _f, _i, _f, _f, _i, _f, _f, _f, _i, _i, _i, _i, _f, _f,
                                _f, ctypes.c_size_t, _f, _f, _f, _i, _f], _i),
    "tpspp_blocked_to_nchw_bf16": ([_f, _i, _i, _i, _f, _f], _i),
    "tpspp_attn_tensor2idx_fwd": ([_f, _i, _i, _i, _i, _i, _f, _f, _f], _i),
}

_lib = None


class TpsppError(RuntimeError):
    pass


def exported_symbols():
    """Names include/tpspp.h declares (kept in sync by tests/test_capi_symbols.py)."""
    return sorted(_SIGNATURES)


def lib():
    """The loaded library; raises if it is absent (build it: `python -m tps_pp_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TpsppError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m tps_pp_amd.build`). There is no CPU or PyTorch fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError if the ABI lost a symbol
            fn.argtypes = argtypes
            fn.restype = restype
        got = L.tpspp_abi_version()
        if got != ABI_VERSION:
            raise TpsppError(f"libtpspp_hip.so ABI {got} != binding ABI {ABI_VERSION}: rebuild")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().tpspp_last_error().decode("utf-8", "replace")
        raise TpsppError(f"{what} failed ({rc}): {msg}")
