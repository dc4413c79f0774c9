"""Loads liblidk.so and declares the C-ABI of include/lidk.h for ctypes."""
import ctypes as C
import os

import torch

F32, BF16 = 0, 1
ACT_NONE, ACT_SWISH, ACT_RELU, ACT_SWISH_GRAD, ACT_GELU, ACT_GELU_GRAD = 0, 1, 2, 3, 4, 5
LN_PARTIAL_BLOCKS = 256
LN_BWD_BLOCKS = 1024
BN_PARTIAL_BLOCKS = 1024
OPT_CHUNK = 8192
ERR_UNSUPPORTED = -3


class LidkError(RuntimeError):
    pass


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblidk.so")


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("lda", C.c_int), ("ldb", C.c_int),
                ("bias", C.c_void_p), ("act", C.c_int), ("alpha", C.c_float),
                ("res", C.c_void_p), ("ldres", C.c_int),
                ("out", C.c_void_p), ("ldo", C.c_int), ("out_f32", C.c_int),
                ("out2", C.c_void_p), ("ldo2", C.c_int),
                ("aux", C.c_void_p), ("ldaux", C.c_int),
                ("splitk", C.c_int)]


_P, _I, _L, _F, _D, _U64 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_uint64

# name -> (restype, argtypes); must list every function declared in include/lidk.h (tests/test_abi.py checks)
SIGNATURES = {
    "lidk_version": (_I, []),
    "lidk_normalize_wav": (_I, [_P, _P, _I, _I, _P, _P]),
    "lidk_dither_preemph": (_I, [_P, _P, _P, _I, _I, _F, _F, _U64, _P]),
    "lidk_speed_perturb": (_I, [_P, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P]),
    "lidk_logmel": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _F, _P, _P, _P, _P]),
    "lidk_wav2mel": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _F, _P, _F, _F, _U64, _P, _P, _P, _P]),
    "lidk_scale_cast": (_I, [_P, _I, _P, _I, _L, _F, _P]),
    "lidk_scale_cast_2d": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _F, _P]),
    "lidk_dropout": (_I, [_P, _I, _P, _I, _P, _P, _L, _F, _U64, _P]),
    "lidk_dropout_add": (_I, [_P, _P, _P, _P, _L, _F, _U64, _P]),
    "lidk_relu_bwd": (_I, [_P, _P, _P, _L, _I, _P]),
    "lidk_colsum": (_I, [_P, _I, _I, _P, _P, _I, _I, _F, _P]),
    "lidk_transpose": (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "lidk_reduce_partials_f64": (_I, [_P, _I, _I, _P, _P, _D, _P]),
    "lidk_layernorm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P]),
    "lidk_layernorm_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _P]),
    "lidk_layernorm_param_grads": (_I, [_P, _I, _I, _P, _P, _P]),
    "lidk_layernorm_param_grads_rows": (_I, [_P, _I, _I, _P, _P, _P]),
    "lidk_ln_param_grads_desc_bytes": (_I, []),
    "lidk_layernorm_param_grads_grouped": (_I, [_P, _I, _I, _P]),
    "lidk_layernorm2_fwd": (_I, [_P] * 11 + [_I, _I, _F, _I, _P]),
    "lidk_layernorm2_bwd": (_I, [_P] * 12 + [_F, _P, _P, _I, _I, _I, _P]),
    "lidk_gemm_nt": (_I, [C.POINTER(GemmArgs), _I, _P]),
    "lidk_gemm_option": (_I, [C.c_char_p, _L]),
    "lidk_gemm_nt_bn_sums": (_I, [C.POINTER(GemmArgs), _P, _P, _P, _P, _P, C.POINTER(C.c_int), _I, _P]),
    "lidk_ln_gemm_supported": (_I, [_I, _I, _I, _I]),
    "lidk_ln_gemm_nt": (_I, [_P, _P, _I, _P, _P, _F, _P, _P, _P, _I, _P]),
    "lidk_ffn_fwd_supported": (_I, [_I, _I, _I, _I]),
    "lidk_ffn_fwd": (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P]),
    "lidk_ffn_fwd_ln": (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F] + [_P] * 11 + [_I, _I, _I, _I, _P]),
    "lidk_ffn_bwd_ln2": (_I, [_P, _P, _P, _I, _P, _I, _P] + [_P] * 9 + [_P, _P, _F, _P, _P, _I, _I, _I, _I, _P]),
    "lidk_ffn_bwd_partial_rows": (_I, [_I]),
    "lidk_ffn_option": (_I, [C.c_char_p, _L]),
    "lidk_mt_chunk_bytes": (_I, []),
    "lidk_mt_chunk_elems": (_I, []),
    "lidk_adam_multi": (_I, [_P, _I, _F, _F, _F, _F, _F, _D, _D, _I, _P]),
    "lidk_sgd_multi": (_I, [_P, _I, _F, _F, _F, _F, _I, _I, _I, _P]),
    "lidk_cast_transpose_desc_bytes": (_I, []),
    "lidk_cast_transpose_grouped": (_I, [_P, _I, _I, _P]),
    "lidk_ffn_bwd": (_I, [_P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _P]),
    "lidk_dgrad_ln_bwd_supported": (_I, [_I, _I, _I, _I]),
    "lidk_dgrad_ln_bwd": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _P]),
    "lidk_gemm_tn": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _F, _I, _I, _P]),
    "lidk_gemm_tn_desc_bytes": (_I, []),
    "lidk_gemm_tn_grouped": (_I, [_P, _I, _I, _I, _P]),
    "lidk_gemm_tn_grouped128": (_I, [_P, _I, _I, _P]),
    "lidk_gemm_tn_grouped256": (_I, [_P, _I, _I, _P]),
    "lidk_attn_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_attn_bwd_relpos_supported": (_I, [_I, _I, _I]),
    "lidk_attn_recompute_supported": (_I, [_I, _I, _I]),
    "lidk_attn_bwd_relpos": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_attn_bwd": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_attn_max_frames": (_I, [_I, _I]),
    "lidk_attn_ldp": (_I, [_I, _I, _I]),
    "lidk_selftest_tr16": (_I, [_P, _P, _P]),
    "lidk_glu_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "lidk_glu_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "lidk_glu_dwconv_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_dwconv_bwd_input_glu": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_dwconv_bwd_input_bn_glu": (_I, [_P, _P, _P, _P, _P, _P, _P, _D, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_dwconv_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_dwconv_stat_parts": (_I, [_I, _I, _I, _I]),
    "lidk_dwconv_bwd_weight_bn_supported": (_I, [_I, _I]),
    "lidk_dwconv_bwd_weight_bn": (_I, [_P] * 8 + [_D, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_dwconv_bwd_input": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_dwconv_bwd_weight": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_bn_train_stats_from_partials": (_I, [_P, _I, _D, _P, _P, _P, _P, _P, _F, _F, _I, _P]),
    "lidk_bn_train_stats": (_I, [_P, _D, _P, _P, _P, _P, _P, _F, _F, _I, _P]),
    "lidk_bn_eval_stats": (_I, [_P, _P, _P, _P, _F, _I, _P]),
    "lidk_bn_swish_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "lidk_bn_swish_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "lidk_bn_swish_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _D, _P, _P, _P, _I, _I, _I, _P]),
    "lidk_im2col_k3s2": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "lidk_ctc_workspace_bytes": (_L, [_I, _I, _I, _I]),
    "lidk_ctc_loss": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    "lidk_lid_score": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "lidk_wav_layernorm": (_I, [_P, _P, _I, _I, _P, _F, _P]),
    "lidk_conv0_ln_fwd": (_I, [_P, _I, _I, _P, _P, _P, _P, _F, _P, _I, _I, _I, _P]),
    "lidk_ln_gelu_fwd": (_I, [_P, _P, _P, _P, _L, _I, _F, _I, _P]),
    "lidk_ln_gelu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "lidk_conv0_ln_bwd": (_I, [_P, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "lidk_wavlm_conv0_workspace": (_L, [_I, _I, _I]),
    "lidk_wavlm_conv0": (_I, [_P, _I, _I, _P, _P, _P, _F, _P, _I, _I, _I, _P, _P]),
    "lidk_wavlm_posconv_prep": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _L, _P]),
    "lidk_wavlm_add_rows": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "lidk_wavlm_apply_mask": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "lidk_wavlm_gate": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "lidk_wavlm_attn_max_frames": (_I, [_I]),
    "lidk_wavlm_attn_ldp": (_I, [_I]),
    "lidk_wavlm_attn_probs": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lidk_wavlm_attn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lidk_wavlm_gate_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "lidk_wavlm_posconv_dprep": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _L, _P]),
    "lidk_wavlm_attn_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lidk_wavlm_attn_fwd_probs": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lidk_wavlm_attn_bias_grads": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lidk_wavlm_conv_dlast": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "lidk_wavlm_conv_col2im": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_wavlm_conv0_bwd": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "lidk_zero_padded_rows": (_I, [_P, _P, _I, _I, _I, _P]),
    "lidk_hidden_mix_axpy": (_I, [_P, _P, _I, _I, _P, _L, _I, _P]),
    "lidk_hidden_mix_dot": (_I, [_P, _P, _P, _L, _P]),
    "lidk_hidden_mix_wgrad": (_I, [_P, _P, _P, _I, _P]),
    "lidk_xattn_max_frames": (_I, [_I]),
    "lidk_xattn_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _U64, _I, _I, _I, _I, _I, _P]),
    "lidk_xattn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _F, _U64, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lidk_ctc_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lidk_ctc_backward": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P, _P]),
    "lidk_ctc_greedy": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "lidk_lid_mlp": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "lidk_novograd_step": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _I, _F, _P, _P, _P]),
    "lidk_cast_weights": (_I, [_P, _P, _P, _I, _L, _I, _P]),
}

_lib = None


def lib():
    """The loaded library; raises LidkError (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise LidkError(f"{path} not found: build it with `make -C speech-lid_amd/csrc` "
                            "(or __graft_entry__.build()); there is no CPU fallback")
        try:
            handle = C.CDLL(path)
        except OSError as e:          # e.g. no ROCm runtime on this machine
            raise LidkError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise LidkError(f"unsupported activation dtype {dt}")


def check(rc: int, what: str):
    if rc != 0:
        names = {-1: "LIDK_ERR_ARG", -2: "LIDK_ERR_LAUNCH", -3: "LIDK_ERR_UNSUPPORTED"}
        raise LidkError(f"{what} failed: {names.get(rc, rc)}")
