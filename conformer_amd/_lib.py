"""ctypes binding of libconformer_hip.so (the C ABI declared in include/conformer_hip.h).

The product path has NO fallback: if the shared library is missing, or a call returns a negative status,
an exception is raised.  Nothing here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CONFORMER_AMD_LIB: A/B a differently built library from tools/*, development only)
LIB_PATH = os.environ.get("CONFORMER_AMD_LIB") or os.path.join(_HERE, "lib", "libconformer_hip.so")

_P, _I, _L, _F, _U = c_void_p, c_int, c_int64, c_float, c_uint64
ABI_VERSION = 4        # CFM_ABI_VERSION of include/conformer_hip.h (checked in load())

# name -> (restype, argtypes).  Mirrors include/conformer_hip.h one to one (checked by tests/test_abi.py).
SIGNATURES = {
    "cfm_version": (c_int, []),
    "cfm_abi_version": (c_int, []),
    "cfm_gemm_bias_stats_f32": (c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _L, _L, _P]),
    "cfm_gemm_bias_residual_stats_f32": (c_int, [_P, _P, _P, _P, _F, _P, _P, _L, _I, _I, _L, _L, _L, _P]),
    "cfm_gemm_splitk_f32": (c_int, [_I, _P, _P, _P, _P, _F, _P, _P, _I, _L, _I, _I, _L, _L, _L, _P]),
    "cfm_gemm_lnfold_f32": (c_int, [_I, _P, _P, _I, _F, _P, _P, _P, _P, _L, _I, _I, _L, _L, _P]),
    "cfm_layernorm_fwd_stats_f32": (c_int, [_P, _P, _P, _P, _P, _L, _I, _F, _P]),
    "cfm_ffn_pack_elems": (ctypes.c_int64, [_I, _I]),
    "cfm_ffn_pack_f32": (c_int, [_P, _P, _P, _I, _I, _P]),
    "cfm_rowgemm_pack_f32": (c_int, [_P, _P, _I, _I, _I, _P]),
    "cfm_rowchain_f32": (c_int, [_I, _I, _I, _I, _P, _L, _P, _P, _P, _L, _P, _L, _P, _I, _F, _P, _P, _P, _P, _F, _I, _P, _L, _P, _P, _P, _F,
                                 _P, _P, _P, _F, _P, _L, _L, _I, _P]),
    "cfm_ffn_tile_stride_f4": (ctypes.c_int64, [_I]),
    "cfm_ffn_rotate": (c_int, []),
    "cfm_ffn_fused_f32": (c_int, [_P, _L, _P, _I, _F, _P, _P, _P, _P, _F, _P, _L, _I, _P, _P, _P, _F, _L, _I, _I, _P]),
    "cfm_strerror": (c_char_p, [_I]),
    "cfm_device_check": (c_int, []),
    "cfm_subsampled_length": (c_int64, [_L]),
    "cfm_subsampled_lengths_i64": (c_int, [_P, _P, _I, _P]),
    "cfm_layernorm_fwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _F, _P]),
    "cfm_gemm_bias_f32": (c_int, [_P, _P, _P, _P, _L, _I, _I, _L, _L, _P]),
    "cfm_gemm_bias_swish_f32": (c_int, [_P, _P, _P, _P, _L, _I, _I, _L, _L, _P]),
    "cfm_gemm_bias_relu_f32": (c_int, [_P, _P, _P, _P, _L, _I, _I, _L, _L, _P]),
    "cfm_gemm_bias_glu_f32": (c_int, [_P, _P, _P, _P, _L, _I, _I, _L, _L, _P]),
    "cfm_gemm_bias_residual_f32": (c_int, [_P, _P, _P, _P, _F, _P, _L, _I, _I, _L, _L, _L, _P]),
    "cfm_gemm_mfma16_f32": (c_int, [_I, _I, _P, _I, _P, _I, _P, _P, _F, _P, _I, _P, _I, _L, _I, _I, _L, _L, _L, _F, _U, _P]),
    "cfm_layernorm_fwd_out16_f32": (c_int, [_I, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P]),
    "cfm_cast16_f32": (c_int, [_I, _P, _P, _L, _P]),
    "cfm_cast16_multi_f32": (c_int, [_I, _P, _I, _P]),
    "cfm_relpos_attention_mfma16_f32": (c_int, [_I, _P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _F, _U, _P]),
    "cfm_relpos_attention_rows_mfma16_f32": (c_int, [_I, _P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "cfm_relpos_attention_io16_mfma16_f32": (c_int, [_I, _P, _P, _P, _I, _L, _P, _L, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _P]),
    "cfm_subsample_conv2_relu_mfma16_f32": (c_int, [_I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfm_subsample_conv1_relu_out16_f32": (c_int, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_gemm_bwd_batched_mfma16_f32": (c_int, [_I, _P, _I, _L, _P, _I, _I, _L, _P, _I, _L, _F, _P, _L, _I, _I, _I, _L, _I, _I, _I, _I,
                                                _L, _L, _L, _L, _L, _L, _F, _U, _I, _P]),
    "cfm_subsample_conv2_bwd_weight_mfma16_f32": (c_int, [_I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_subsample_conv2_bwd_input_mfma16_f32": (c_int, [_I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_subsample_conv2_rowtab_elems": (ctypes.c_int64, [_I, _I, _I]),
    "cfm_subsample_conv2_bwd_weight_h16_mfma16_f32": (c_int, [_I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_relu_bwd_out16_f32": (c_int, [_I, _P, _P, _P, _L, _P]),
    "cfm_dwconv_bn_swish_fwd_out16_f32": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _P]),
    "cfm_glu_bwd_out16_f32": (c_int, [_I, _P, _P, _P, _L, _I, _P]),
    "cfm_subsample_conv2_bwd_input_fwdkernel_mfma16_f32": (c_int, [_I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_subsample_conv2_bwd_input_fwdkernel_out16_mfma16_f32": (c_int, [_I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_subsample_conv1_bwd_d16_f32": (c_int, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_reflect_pad_f32": (c_int, [_P, _P, _I, _L, _I, _L, _P]),
    "cfm_power_mel_log_f32": (c_int, [_P, _L, _P, _P, _I, _I, _I, _I, _F, _P]),
    "cfm_power_mel_log_mfma_f32": (c_int, [_P, _L, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "cfm_dft_frames_f32": (c_int, [_P, _P, _P, _L, _I, _I, _I, _P]),
    "cfm_specaugment_apply_f32": (c_int, [_P, _I, _I, _I, _P, _I, _F, _P]),
    "cfm_gemm_bias_swish_save_f32": (c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _L, _L, _P]),
    "cfm_gemm_bwd_f32": (c_int, [_P, _I, _L, _P, _I, _L, _P, _L, _F, _P, _L, _I, _I, _L, _I, _P]),
    "cfm_gemm_bwd_batched_f32": (c_int, [_P, _I, _L, _P, _I, _L, _P, _L, _F, _P, _L, _I, _I, _L, _I, _I, _I, _I,
                                         _L, _L, _L, _L, _L, _L, _F, _U, _P]),
    "cfm_gemm_train_f32": (c_int, [_I, _P, _P, _P, _P, _F, _P, _P, _L, _I, _I, _L, _L, _L, _F, _U, _P]),
    "cfm_relpos_attention_train_f32": (c_int, [_P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _F, _U, _P]),
    "cfm_dropout_f32": (c_int, [_P, _P, _L, _F, _U, _P]),
    "cfm_dropout_out16_f32": (c_int, [_I, _P, _P, _L, _F, _U, _P]),
    "cfm_layernorm_bwd_dx_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "cfm_layernorm_bwd_params_f32": (c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "cfm_layernorm_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, ctypes.c_size_t, _P]),
    "cfm_layernorm_bwd_workspace_bytes": (ctypes.c_size_t, [_L, _I]),
    "cfm_colsum_f32": (c_int, [_P, _L, _L, _I, _F, _P, _P]),
    "cfm_glu_fwd_f32": (c_int, [_P, _P, _L, _I, _P]),
    "cfm_glu_bwd_f32": (c_int, [_P, _P, _P, _L, _I, _P]),
    "cfm_dwconv_bn_swish_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P, _P,
                                            _I, _I, _I, _I, _P]),
    "cfm_dwconv_bn_stats_workspace_bytes": (ctypes.c_size_t, [_I, _I, _I]),
    "cfm_dwconv_bn_stats_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P, ctypes.c_size_t, _P]),
    "cfm_debug_attention_bwd_trace_f32": (c_int, [_P]),
    "cfm_debug_attention_bwd_trace_mfma16": (c_int, [_P]),
    "cfm_debug_gemm_mfma16_trace": (c_int, [_P]),
    "cfm_debug_gemm_mfma16_force_tile": (c_int, [_I]),
    "cfm_relpos_attention_bwd_mfma16_f32": (c_int, [_I, _P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _L, _P, _L,
                                                    _P, _P, _I, _I, _I, _I, _F, _U, _P]),
    "cfm_relpos_attention_bwd_f32": (c_int, [_P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _L, _P, _L, _P, _P,
                                             _I, _I, _I, _I, _F, _U, _I, _P]),
    "cfm_relu_bwd_f32": (c_int, [_P, _P, _P, _L, _P]),
    "cfm_subsample_conv2_bwd_weight_f32": (c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_pack_conv2_weight_t_f32": (c_int, [_P, _P, _I, _P]),
    "cfm_subsample_conv2_bwd_input_f32": (c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_subsample_conv1_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_adam_step_f32": (c_int, [_P, _I, _F, _F, _F, _F, _F, _F, _P]),
    "cfm_greedy_ctc_decode_f32": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfm_lstm_fwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfm_lstm_fwd_frag_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfm_lstm_bwd_frag_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfm_debug_lstm_trace": (c_int, [_P]),
    "cfm_debug_dw16_trace": (c_int, [_P]),
    "cfm_linear_bwd_weight_mfma16_f32": (c_int, [_I, _P, _I, _L, _P, _I, _L, _P, _L, _P, _I, _I, _L, _F, _P]),
    "cfm_lstm_fwd_mfma16_f32": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfm_lstm_bwd_mfma16_f32": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfm_lstm_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cfm_swish_bn_eval_f32": (c_int, [_P, _P, _P, _P, _P, _F, _P, _L, _I, _P]),
    "cfm_swish_bn_stats_f32": (c_int, [_P, _P, _P, _P, _P, _F, _L, _I, _P]),
    "cfm_swish_bn_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _F, _I, _P, _P, _P, _L, _I, _P]),
    "cfm_split_pack_elems": (ctypes.c_int64, [_I, _I, _I]),
    "cfm_split_pack_bf16_f32": (c_int, [_I, _P, _P, _I, _I, _P]),
    "cfm_gemm_split_bf16_f32": (c_int, [_I, _I, _P, _P, _P, _P, _F, _P, _L, _I, _I, _L, _L, _L, _P]),
    "cfm_subsample_conv2_relu_split_bf16_f32": (c_int, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_ctc_workspace_floats": (ctypes.c_int64, [_I, _I, _I]),
    "cfm_ctc_loss_fwd_f32": (c_int, [_P, _P, _P, _L, _L, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "cfm_ctc_loss_bwd_f32": (c_int, [_P, _P, _P, _L, _L, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "cfm_relpos_attention_rows_f32": (c_int, [_P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "cfm_debug_attention_trace_f32": (c_int, [_P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P]),
    "cfm_debug_set_bwd_tile": (c_int, [_I]),
    "cfm_debug_set_attention_waves": (c_int, [_I]),
    "cfm_debug_gemm_cfg_f32": (c_int, [_I, _P, _P, _P, _P, _F, _P, _L, _I, _I, _P, _P]),
    "cfm_debug_set_conv2_bk": (c_int, [_I]),
    "cfm_debug_ffn_trace": (c_int, [_P, _I]),
    "cfm_debug_ffn_variant": (c_int, [_I]),
    "cfm_debug_ffn_layout": (c_int, [_I, _I]),
    "cfm_relpos_table_f32": (c_int, [_P, _P, _I, _I, _P]),
    "cfm_relpos_attention_fwd_f32": (c_int, [_P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _P]),
    "cfm_dwconv_bn_swish_fwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _P]),
    "cfm_subsample_conv1_relu_f32": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_pack_conv2_weight_f32": (c_int, [_P, _P, _I, _P]),
    "cfm_subsample_conv2_relu_f32": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfm_pack_linear_weight_f32": (c_int, [_P, _P, _I, _I, _I, _P]),
}


class ConformerHipError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Load the library once.  `import torch` first so the HIP runtime torch ships is the one bound."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ConformerHipError(
            f"{LIB_PATH} is missing: build it with `python -m conformer_amd.build` "
            "(there is no CPU/eager fallback for the Conformer hot path)")
    import torch  # noqa: F401  (loads libamdhip64 from torch/lib before our DT_NEEDED is resolved)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    # a library built against another revision of include/conformer_hip.h would take shifted arguments silently (ctypes checks
    # nothing): refuse it before the first call (matters most for the CONFORMER_AMD_LIB override)
    try:
        lib.cfm_abi_version.restype = c_int
        got = int(lib.cfm_abi_version())
    except AttributeError:
        got = 0
    if got != ABI_VERSION:
        raise ConformerHipError(f"{LIB_PATH} implements C-ABI revision {got}, this binding needs {ABI_VERSION}: rebuild it with "
                                "`python -m conformer_amd.build`")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class CastItem(ctypes.Structure):
    """cfm_cast_item of include/conformer_hip.h"""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("rows", ctypes.c_int64), ("cols", ctypes.c_int64),
                ("transpose", ctypes.c_int)]


CALLS = [0]          # C-ABI calls checked so far (bench.py reports the count of one forward: one call = one kernel launch there)


def check(status: int, what: str) -> None:
    CALLS[0] += 1
    if status != 0:
        msg = load().cfm_strerror(status).decode()
        raise ConformerHipError(f"{what} failed: {msg} (status {status})")
