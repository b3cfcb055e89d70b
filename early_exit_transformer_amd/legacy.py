"""Drop-in mirror of the reference's legacy early-exit transformer encoder.

Same constructor keywords, ``forward(src)`` signature, return shape and state_dict keys as
``Early_encoder`` (/root/reference/models/model/early_exit.py:497-562) and its parts
``Encoder`` (models/model/encoder.py:13-37), ``EncoderLayer`` (models/blocks/encoder_layer.py:14-44),
``MultiHeadAttention`` (models/layers/multi_head_attention.py:11-68) and ``PositionwiseFeedForward``
(models/layers/position_wise_feed_forward.py:9-23).  The classes hold parameters only; the arithmetic
(pre-norm attention without mask, ReLU feed-forward, group-final LayerNorm, exit heads) runs in
libeec.so on the same kernels as the Conformer path.  Unlike the Conformer's torchaudio layers this
code IS in the reference tree, so tests compare against the reference's own, unmodified classes.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor, nn

from . import capi
from .conformer import _HipOnly
from .model import Conv1dSubampling, PositionalEncoding, _HipEncoderMixin


class MultiHeadAttention(_HipOnly):
    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.n_head = n_head
        self.w_q = nn.Linear(d_model, d_model)
        self.w_k = nn.Linear(d_model, d_model)
        self.w_v = nn.Linear(d_model, d_model)
        self.w_concat = nn.Linear(d_model, d_model)


class PositionwiseFeedForward(_HipOnly):
    def __init__(self, d_model: int, hidden: int, drop_prob: float = 0.1):
        super().__init__()
        self.linear1 = nn.Linear(d_model, hidden)
        self.linear2 = nn.Linear(hidden, d_model)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(p=drop_prob)


class EncoderLayer(_HipOnly):
    def __init__(self, d_model: int, ffn_hidden: int, n_head: int, drop_prob: float):
        super().__init__()
        self.attention = MultiHeadAttention(d_model=d_model, n_head=n_head)
        self.norm1 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(p=drop_prob)
        self.ffn = PositionwiseFeedForward(d_model=d_model, hidden=ffn_hidden, drop_prob=drop_prob)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout2 = nn.Dropout(p=drop_prob)


class Encoder(_HipOnly):
    def __init__(self, enc_voc_size, max_len, d_model, ffn_hidden, n_head, n_layers, drop_prob, device=None):
        super().__init__()
        self.layers = nn.ModuleList([EncoderLayer(d_model=d_model, ffn_hidden=ffn_hidden, n_head=n_head,
                                                  drop_prob=drop_prob) for _ in range(n_layers)])
        self.layer_norm = nn.LayerNorm(d_model)


_LEGACY_KEYS = {
    "norm1_w": "norm1.weight", "norm1_b": "norm1.bias",
    "wq": "attention.w_q.weight", "bq": "attention.w_q.bias", "wk": "attention.w_k.weight", "bk": "attention.w_k.bias",
    "wv": "attention.w_v.weight", "bv": "attention.w_v.bias", "wo": "attention.w_concat.weight",
    "bo": "attention.w_concat.bias", "norm2_w": "norm2.weight", "norm2_b": "norm2.bias",
    "w1": "ffn.linear1.weight", "b1": "ffn.linear1.bias", "w2": "ffn.linear2.weight", "b2": "ffn.linear2.bias",
}


class EecLegacyLayerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LEGACY_KEYS]


class EecLegacyParams(C.Structure):
    _fields_ = [("sub0_w", C.c_void_p), ("sub0_b", C.c_void_p), ("sub1_w", C.c_void_p), ("sub1_b", C.c_void_p),
                ("pe", C.c_void_p), ("layers", C.POINTER(EecLegacyLayerParams)),
                ("group_ln_w", C.POINTER(C.c_void_p)), ("group_ln_b", C.POINTER(C.c_void_p)),
                ("head_w", C.POINTER(C.c_void_p)), ("head_b", C.POINTER(C.c_void_p))]


class Early_encoder(_HipEncoderMixin, nn.Module):
    """``forward(src[B, n_mels, T]) -> [E, B, T', V]`` log-probs (no lengths: the reference passes mask=None)."""

    def __init__(self, src_pad_idx, n_enc_exits, enc_voc_size, dec_voc_size, d_model, n_head, max_len, d_feed_forward,
                 n_enc_layers, features_length, drop_prob, device=None):
        nn.Module.__init__(self)
        self.src_pad_idx, self.n_enc_exits, self.device = src_pad_idx, n_enc_exits, device
        self.conv_subsample = Conv1dSubampling(features_length, d_model)
        self.positional_encoder = PositionalEncoding(d_model, drop_prob, max_len)
        self.linears = nn.ModuleList([nn.Linear(d_model, dec_voc_size) for _ in range(n_enc_exits)])
        self.encoders = nn.ModuleList([
            Encoder(d_model=d_model, n_head=n_head, max_len=max_len, ffn_hidden=d_feed_forward,
                    enc_voc_size=enc_voc_size, drop_prob=drop_prob, n_layers=n_enc_layers, device=device)
            for _ in range(n_enc_exits)])
        self._hip_init(d_model, n_head, d_feed_forward, 1, n_enc_exits, n_enc_layers, features_length, dec_voc_size, max_len)
        self._cfg.arch = capi.ARCH_LEGACY

    def _param_tensors(self):
        return list(self.conv_subsample.parameters()) + list(self.encoders.parameters()) + \
            list(self.linears.parameters()) + [self.positional_encoder.pe]

    def _pack(self, lib, device, sd, ptr) -> None:
        E, L = self._cfg.n_exits, self._cfg.layers_per_exit
        layers = (EecLegacyLayerParams * (E * L))()
        for e in range(E):
            for l in range(L):
                for field, suffix in _LEGACY_KEYS.items():
                    setattr(layers[e * L + l], field, ptr(f"encoders.{e}.layers.{l}.{suffix}"))
        arr = lambda fmt: (C.c_void_p * E)(*[ptr(fmt.format(e=e)) for e in range(E)])
        params = EecLegacyParams(ptr("conv_subsample.sequential.0.weight"), ptr("conv_subsample.sequential.0.bias"),
                                 ptr("conv_subsample.sequential.1.weight"), ptr("conv_subsample.sequential.1.bias"),
                                 ptr("positional_encoder.pe"), layers, arr("encoders.{e}.layer_norm.weight"),
                                 arr("encoders.{e}.layer_norm.bias"), arr("linears.{e}.weight"), arr("linears.{e}.bias"))
        stream = torch.cuda.current_stream(device).cuda_stream
        capi.check(lib.eec_encoder_pack_legacy(self._enc, C.byref(params), C.c_void_p(stream)), "eec_encoder_pack_legacy")

    def forward(self, src: Tensor) -> Tensor:
        lengths = torch.full((src.size(0),), src.size(2), dtype=torch.int64)
        return self._run_encoder(src, lengths)[0]
