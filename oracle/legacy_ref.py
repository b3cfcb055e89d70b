"""fp32 torch-CPU restatement of the reference's legacy early-exit transformer encoder.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Restates, with the same module tree (state_dict keys):
``Early_encoder`` /root/reference/models/model/early_exit.py:497-562, ``Encoder`` models/model/encoder.py:13-37,
``EncoderLayer`` models/blocks/encoder_layer.py:14-44, ``MultiHeadAttention`` models/layers/multi_head_attention.py:11-68
(+ ``ScaleDotProductAttention`` models/layers/scale_dot_product_attention.py:11-42, mask=None path),
``PositionwiseFeedForward`` models/layers/position_wise_feed_forward.py:9-23.
All of these are importable from the reference tree; tests/test_oracle.py checks bit-equality with them.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .conformer_ref import SinusoidPE, Subsample


class MultiHeadAttention(nn.Module):
    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.n_head = n_head
        self.w_q, self.w_k, self.w_v, self.w_concat = (nn.Linear(d_model, d_model) for _ in range(4))

    def forward(self, x: Tensor) -> Tensor:
        b, t, d = x.shape
        dh = d // self.n_head
        split = lambda y: y.view(b, t, self.n_head, dh).transpose(1, 2)
        q, k, v = split(self.w_q(x)), split(self.w_k(x)), split(self.w_v(x))
        score = torch.softmax((q @ k.transpose(2, 3)) / math.sqrt(dh), dim=-1)
        return self.w_concat((score @ v).transpose(1, 2).contiguous().view(b, t, d))


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_model: int, hidden: int, drop_prob: float = 0.1):
        super().__init__()
        self.linear1, self.linear2 = nn.Linear(d_model, hidden), nn.Linear(hidden, d_model)
        self.relu, self.dropout = nn.ReLU(), nn.Dropout(p=drop_prob)

    def forward(self, x: Tensor) -> Tensor:
        return self.linear2(self.dropout(self.relu(self.linear1(x))))


class EncoderLayer(nn.Module):
    def __init__(self, d_model, ffn_hidden, n_head, drop_prob):
        super().__init__()
        self.attention = MultiHeadAttention(d_model, n_head)
        self.norm1, self.dropout1 = nn.LayerNorm(d_model), nn.Dropout(p=drop_prob)
        self.ffn = PositionwiseFeedForward(d_model, ffn_hidden, drop_prob)
        self.norm2, self.dropout2 = nn.LayerNorm(d_model), nn.Dropout(p=drop_prob)

    def forward(self, x: Tensor) -> Tensor:
        x = self.dropout1(self.attention(self.norm1(x))) + x
        return self.dropout2(self.ffn(self.norm2(x))) + x


class Encoder(nn.Module):
    def __init__(self, d_model, ffn_hidden, n_head, n_layers, drop_prob):
        super().__init__()
        self.layers = nn.ModuleList([EncoderLayer(d_model, ffn_hidden, n_head, drop_prob) for _ in range(n_layers)])
        self.layer_norm = nn.LayerNorm(d_model)

    def forward(self, x: Tensor) -> Tensor:
        for layer in self.layers:
            x = layer(x)
        return self.layer_norm(x)


class EarlyEncoderRef(nn.Module):
    def __init__(self, src_pad_idx, n_enc_exits, enc_voc_size, dec_voc_size, d_model, n_head, max_len, d_feed_forward,
                 n_enc_layers, features_length, drop_prob, device="cpu"):
        super().__init__()
        self.conv_subsample = Subsample(features_length, d_model)
        self.positional_encoder = SinusoidPE(d_model, drop_prob, max_len)
        self.linears = nn.ModuleList([nn.Linear(d_model, dec_voc_size) for _ in range(n_enc_exits)])
        self.encoders = nn.ModuleList([Encoder(d_model, d_feed_forward, n_head, n_enc_layers, drop_prob)
                                       for _ in range(n_enc_exits)])

    def forward(self, src: Tensor) -> Tensor:
        enc = self.positional_encoder(self.conv_subsample(src).permute(0, 2, 1))
        outs = []
        for head, encoder in zip(self.linears, self.encoders):
            enc = encoder(enc)
            outs.append(F.log_softmax(head(enc), dim=2).unsqueeze(0))
        return torch.cat(outs)
