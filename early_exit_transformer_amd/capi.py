"""ctypes binding of libeec.so (include/eec.h).  There is no CPU fallback: if the
library is missing, loading raises and every product entry point fails loudly."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .build import LIB_PATH

PREC_F16X3, PREC_MIXED, PREC_F16, PREC_F16F8 = 0, 1, 2, 3
ARCH_CONFORMER, ARCH_LEGACY = 0, 1
PRECISIONS = {"f16x3": PREC_F16X3, "mixed": PREC_MIXED, "f16": PREC_F16, "f16f8": PREC_F16F8}

_LAYER_FIELDS = [
    "ffn1_ln_w", "ffn1_ln_b", "ffn1_w1", "ffn1_b1", "ffn1_w2", "ffn1_b2",
    "attn_ln_w", "attn_ln_b", "attn_in_w", "attn_in_b", "attn_out_w", "attn_out_b",
    "conv_ln_w", "conv_ln_b", "conv_pw1_w", "conv_pw1_b", "conv_dw_w", "conv_dw_b",
    "conv_bn_w", "conv_bn_b", "conv_bn_rm", "conv_bn_rv", "conv_pw2_w", "conv_pw2_b",
    "ffn2_ln_w", "ffn2_ln_b", "ffn2_w1", "ffn2_b1", "ffn2_w2", "ffn2_b2",
    "final_ln_w", "final_ln_b",
]

# state_dict key suffix (inside conformer.{e}.conformer_layers.{l}.) of every eec_layer_params field
LAYER_KEYS = {
    "ffn1_ln_w": "ffn1.sequential.0.weight", "ffn1_ln_b": "ffn1.sequential.0.bias",
    "ffn1_w1": "ffn1.sequential.1.weight", "ffn1_b1": "ffn1.sequential.1.bias",
    "ffn1_w2": "ffn1.sequential.4.weight", "ffn1_b2": "ffn1.sequential.4.bias",
    "attn_ln_w": "self_attn_layer_norm.weight", "attn_ln_b": "self_attn_layer_norm.bias",
    "attn_in_w": "self_attn.in_proj_weight", "attn_in_b": "self_attn.in_proj_bias",
    "attn_out_w": "self_attn.out_proj.weight", "attn_out_b": "self_attn.out_proj.bias",
    "conv_ln_w": "conv_module.layer_norm.weight", "conv_ln_b": "conv_module.layer_norm.bias",
    "conv_pw1_w": "conv_module.sequential.0.weight", "conv_pw1_b": "conv_module.sequential.0.bias",
    "conv_dw_w": "conv_module.sequential.2.weight", "conv_dw_b": "conv_module.sequential.2.bias",
    "conv_bn_w": "conv_module.sequential.3.weight", "conv_bn_b": "conv_module.sequential.3.bias",
    "conv_bn_rm": "conv_module.sequential.3.running_mean", "conv_bn_rv": "conv_module.sequential.3.running_var",
    "conv_pw2_w": "conv_module.sequential.5.weight", "conv_pw2_b": "conv_module.sequential.5.bias",
    "ffn2_ln_w": "ffn2.sequential.0.weight", "ffn2_ln_b": "ffn2.sequential.0.bias",
    "ffn2_w1": "ffn2.sequential.1.weight", "ffn2_b1": "ffn2.sequential.1.bias",
    "ffn2_w2": "ffn2.sequential.4.weight", "ffn2_b2": "ffn2.sequential.4.bias",
    "final_ln_w": "final_layer_norm.weight", "final_ln_b": "final_layer_norm.bias",
}


class EecConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d_model", "n_heads", "d_ff", "dw_kernel", "n_exits",
                                         "layers_per_exit", "n_mels", "vocab", "max_len", "arch")]


class EecLayerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LAYER_FIELDS]


class EecParams(C.Structure):
    _fields_ = [("sub0_w", C.c_void_p), ("sub0_b", C.c_void_p), ("sub1_w", C.c_void_p), ("sub1_b", C.c_void_p),
                ("pe", C.c_void_p), ("layers", C.POINTER(EecLayerParams)),
                ("head_w", C.POINTER(C.c_void_p)), ("head_b", C.POINTER(C.c_void_p))]


class EecDecoderLayerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "ca_in_w", "ca_in_b", "ca_out_w", "ca_out_b",
                                          "w1", "b1", "w2", "b2", "norm1_w", "norm1_b", "norm2_w", "norm2_b", "norm3_w", "norm3_b")]


# field -> state_dict key suffix below ``decoders.{e}.layers.{l}.``
DECODER_LAYER_KEYS = {"sa_in_w": "self_attn.in_proj_weight", "sa_in_b": "self_attn.in_proj_bias",
                      "sa_out_w": "self_attn.out_proj.weight", "sa_out_b": "self_attn.out_proj.bias",
                      "ca_in_w": "multihead_attn.in_proj_weight", "ca_in_b": "multihead_attn.in_proj_bias",
                      "ca_out_w": "multihead_attn.out_proj.weight", "ca_out_b": "multihead_attn.out_proj.bias",
                      "w1": "linear1.weight", "b1": "linear1.bias", "w2": "linear2.weight", "b2": "linear2.bias",
                      "norm1_w": "norm1.weight", "norm1_b": "norm1.bias", "norm2_w": "norm2.weight", "norm2_b": "norm2.bias",
                      "norm3_w": "norm3.weight", "norm3_b": "norm3.bias"}


class EecDecoderParams(C.Structure):
    _fields_ = [("emb", C.c_void_p), ("pe", C.c_void_p), ("layers", C.POINTER(EecDecoderLayerParams)), ("n_layers", C.c_int32),
                ("max_len", C.c_int32), ("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("head_w", C.c_void_p), ("head_b", C.c_void_p)]


# host callback of eec_train_backward_ex: (exit group just finished, or -1 after the stem; user pointer)
GROUP_DONE_FN = C.CFUNCTYPE(None, C.c_int, C.c_void_p)

EXPORTS = ["eec_last_error", "eec_abi_version", "eec_out_frames", "eec_encoder_create", "eec_encoder_destroy",
           "eec_encoder_pack", "eec_encoder_workspace_bytes", "eec_encoder_forward", "eec_greedy_ctc",
           "eec_encoder_set_profiling", "eec_encoder_profile_read", "eec_ctc_loss", "eec_encoder_pack_legacy",
           "eec_encoder_forward_prefix", "eec_encoder_group_workspace_bytes", "eec_encoder_group_forward",
           "eec_encoder_head_forward", "eec_encoder_stem1_forward", "eec_encoder_lengths",
           "eec_ctc_backward_workspace_bytes", "eec_ctc_loss_forward", "eec_ctc_loss_backward", "eec_logsoftmax_backward",
           "eec_ctc_beam_workspace_bytes", "eec_ctc_beam_decode", "eec_ctc_beam_decode_ex",
           "eec_frontend_last_error", "eec_frontend_create", "eec_frontend_destroy", "eec_frontend_frames", "eec_frontend_forward",
           "eec_trainer_last_error", "eec_trainer_create", "eec_trainer_destroy", "eec_trainer_workspace_bytes",
           "eec_train_forward", "eec_train_backward", "eec_train_backward_ex", "eec_train_gemm",
           "eec_train_group_workspace_bytes", "eec_train_group_forward", "eec_train_group_backward", "eec_train_stem_workspace_bytes",
           "eec_train_stem_forward", "eec_train_stem_backward", "eec_train_head_forward", "eec_train_head_backward_scratch_floats",
           "eec_train_head_backward",
           "eec_decoder_last_error", "eec_decoder_workspace_bytes", "eec_decoder_forward",
           "eec_decoder_train_last_error", "eec_decoder_train_workspace_bytes", "eec_decoder_train_forward", "eec_decoder_train_backward",
           "eec_decoder_step_last_error", "eec_decoder_step_max_beams", "eec_decoder_cache_bytes", "eec_decoder_begin", "eec_decoder_step", "eec_decoder_step_multi", "eec_upload_i64_max", "eec_upload_i64", "eec_beam_select"]
KERNEL_CLASSES = ["stem", "ffn", "qkv", "attn", "proj_glu", "proj", "dw_pw2", "head", "chain"]

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen libeec.so and declare the prototypes; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP library is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or early_exit_transformer_amd.build.build_library()). There is no CPU fallback.")
    # torch first: its bundled libamdhip64.so carries the soname libeec.so asks for (libamdhip64.so.7), so the loader
    # reuses it.  Loaded the other way round, /opt/rocm's copy comes in as well and one process holds two HIP runtimes
    # (the second to initialise then reports "no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.eec_last_error.restype = C.c_char_p
    lib.eec_abi_version.restype = C.c_int
    lib.eec_out_frames.argtypes = [C.c_int]
    lib.eec_encoder_create.argtypes = [C.POINTER(EecConfig), C.POINTER(C.c_void_p)]
    lib.eec_encoder_destroy.argtypes = [C.c_void_p]
    lib.eec_encoder_destroy.restype = None
    lib.eec_encoder_pack.argtypes = [C.c_void_p, C.POINTER(EecParams), C.c_void_p]
    lib.eec_encoder_pack_legacy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.eec_encoder_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.eec_encoder_workspace_bytes.restype = C.c_size_t
    lib.eec_encoder_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.eec_encoder_forward_prefix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_encoder_group_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.eec_encoder_group_workspace_bytes.restype = C.c_size_t
    lib.eec_encoder_group_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_encoder_head_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.eec_encoder_stem1_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.eec_encoder_lengths.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.eec_greedy_ctc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p]
    lib.eec_ctc_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
    lib.eec_ctc_backward_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.eec_ctc_backward_workspace_bytes.restype = C.c_size_t
    lib.eec_ctc_loss_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.eec_ctc_loss_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.eec_logsoftmax_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.eec_ctc_beam_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.eec_ctc_beam_workspace_bytes.restype = C.c_size_t
    lib.eec_ctc_beam_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
    lib.eec_ctc_beam_decode_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
    lib.eec_frontend_last_error.restype = C.c_char_p
    lib.eec_frontend_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.eec_frontend_destroy.argtypes = [C.c_void_p]
    lib.eec_frontend_destroy.restype = None
    lib.eec_frontend_frames.argtypes = [C.c_int, C.c_int]
    lib.eec_frontend_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.eec_trainer_last_error.restype = C.c_char_p
    lib.eec_trainer_create.argtypes = [C.POINTER(EecConfig), C.POINTER(C.c_void_p)]
    lib.eec_trainer_destroy.argtypes = [C.c_void_p]
    lib.eec_trainer_destroy.restype = None
    lib.eec_trainer_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.eec_trainer_workspace_bytes.restype = C.c_size_t
    lib.eec_train_forward.argtypes = [C.c_void_p, C.POINTER(EecParams), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_train_backward.argtypes = [C.c_void_p, C.POINTER(EecParams), C.POINTER(EecParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_train_backward_ex.argtypes = [C.c_void_p, C.POINTER(EecParams), C.POINTER(EecParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_void_p, GROUP_DONE_FN, C.c_void_p]
    lib.eec_train_group_workspace_bytes.argtypes = [C.POINTER(EecConfig), C.c_int, C.c_int, C.c_int]
    lib.eec_train_group_workspace_bytes.restype = C.c_size_t
    lib.eec_train_group_forward.argtypes = [C.POINTER(EecConfig), C.POINTER(EecLayerParams), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                            C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_train_group_backward.argtypes = [C.POINTER(EecConfig), C.POINTER(EecLayerParams), C.POINTER(EecLayerParams), C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_train_stem_workspace_bytes.argtypes = [C.POINTER(EecConfig), C.c_int, C.c_int, C.c_int]
    lib.eec_train_stem_workspace_bytes.restype = C.c_size_t
    lib.eec_train_stem_forward.argtypes = [C.POINTER(EecConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_train_stem_backward.argtypes = [C.POINTER(EecConfig), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_train_head_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.eec_train_head_backward_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.eec_train_head_backward_scratch_floats.restype = C.c_size_t
    lib.eec_train_head_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.eec_train_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]
    lib.eec_decoder_last_error.restype = C.c_char_p
    lib.eec_decoder_workspace_bytes.argtypes = [C.c_int] * 7
    lib.eec_decoder_workspace_bytes.restype = C.c_size_t
    lib.eec_decoder_forward.argtypes = [C.POINTER(EecDecoderParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_decoder_train_last_error.restype = C.c_char_p
    lib.eec_decoder_train_workspace_bytes.argtypes = [C.c_int] * 8
    lib.eec_decoder_train_workspace_bytes.restype = C.c_size_t
    lib.eec_decoder_train_forward.argtypes = [C.POINTER(EecDecoderParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_size_t, C.c_void_p]
    lib.eec_decoder_train_backward.argtypes = [C.POINTER(EecDecoderParams), C.POINTER(EecDecoderParams), C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_decoder_step_last_error.restype = C.c_char_p
    lib.eec_decoder_cache_bytes.argtypes = [C.c_int] * 7
    lib.eec_decoder_cache_bytes.restype = C.c_size_t
    lib.eec_decoder_begin.argtypes = [C.POINTER(EecDecoderParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_decoder_step.argtypes = [C.POINTER(EecDecoderParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.eec_decoder_step_multi.argtypes = [C.c_int, C.POINTER(C.POINTER(EecDecoderParams)), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]
    lib.eec_beam_select.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.eec_upload_i64.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.eec_encoder_set_profiling.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.eec_encoder_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().eec_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
