"""fp32 torch-CPU restatement of the early-exit Conformer encoder (the oracle).

TEST INFRASTRUCTURE -- see oracle/__init__.py for who may import this.

What is restated, and from where (all paths relative to /root/reference):

* ``Subsample``            <- models/model/early_exit.py:24-48   (Conv1dSubampling)
* ``SinusoidPE``           <- models/embedding/positional_encoding.py:55-73
* ``EarlyConformerRef``    <- models/model/early_exit.py:565-634 (Early_conformer)
* ``FullConformerEncoderRef`` <- models/model/early_exit.py:637-737,764-800
                              (encoder half of full_conformer, ``_encoder_``)
* ``Conformer`` and below  <- torchaudio.models.conformer (third-party, NOT in
  the reference tree, version unpinned; call sites early_exit.py:16,603-615,627).
  Restated from torchaudio's published module tree so that state_dict keys and
  shapes are identical and each primitive is the
  installed torch CPU kernel.
* ``greedy_ctc``           <- util/beam_infer.py:9-24  (GreedyCTCDecoder)
* ``summed_exit_ctc_loss`` <- train.py:53-65,259

The module tree (attribute names, Sequential indices) is what fixes the
state_dict key contract; the arithmetic is written functionally.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn


# --------------------------------------------------------------------------
# torchaudio.models.conformer restatement
# --------------------------------------------------------------------------
def lengths_to_padding_mask(lengths: Tensor) -> Tensor:
    """True where a frame is padding.  Width is max(lengths) (host sync)."""
    width = int(lengths.max().item())
    steps = torch.arange(width, device=lengths.device, dtype=lengths.dtype)
    return steps.unsqueeze(0) >= lengths.unsqueeze(1)


class _FeedForwardModule(nn.Module):
    """LN -> Linear(D,F) -> SiLU -> Drop -> Linear(F,D) -> Drop; keys sequential.{0,1,4}."""

    def __init__(self, input_dim: int, hidden_dim: int, dropout: float = 0.0):
        super().__init__()
        self.sequential = nn.Sequential(
            nn.LayerNorm(input_dim),
            nn.Linear(input_dim, hidden_dim, bias=True),
            nn.SiLU(),
            nn.Dropout(dropout),
            nn.Linear(hidden_dim, input_dim, bias=True),
            nn.Dropout(dropout),
        )

    def forward(self, x: Tensor) -> Tensor:
        return self.sequential(x)


class _ConvolutionModule(nn.Module):
    """LN -> pw(D,2D) -> GLU(ch) -> dw(K, groups=D) -> BN -> SiLU -> pw(D,D) -> Drop.

    Keys: layer_norm, sequential.{0,2,3,5}.  Runs channel-first; zero padding
    only at the two ends of the padded sequence (padded frames leak into valid
    ones within (K-1)/2 steps -- reproduced, not fixed).
    """

    def __init__(self, input_dim: int, num_channels: int, depthwise_kernel_size: int,
                 dropout: float = 0.0, bias: bool = False):
        super().__init__()
        if (depthwise_kernel_size - 1) % 2 != 0:
            raise ValueError("depthwise_kernel_size must be odd to achieve 'SAME' padding.")
        self.layer_norm = nn.LayerNorm(input_dim)
        self.sequential = nn.Sequential(
            nn.Conv1d(input_dim, 2 * num_channels, 1, stride=1, padding=0, bias=bias),
            nn.GLU(dim=1),
            nn.Conv1d(num_channels, num_channels, depthwise_kernel_size, stride=1,
                      padding=(depthwise_kernel_size - 1) // 2, groups=num_channels, bias=bias),
            nn.BatchNorm1d(num_channels),
            nn.SiLU(),
            nn.Conv1d(num_channels, input_dim, 1, stride=1, padding=0, bias=bias),
            nn.Dropout(dropout),
        )

    def forward(self, x: Tensor) -> Tensor:  # x: [B, T, D]
        y = self.layer_norm(x).transpose(1, 2)
        return self.sequential(y).transpose(1, 2)


class ConformerLayer(nn.Module):
    """ffn1(half) -> MHSA -> conv -> ffn2(half) -> LN, convolution_first=False."""

    def __init__(self, input_dim: int, ffn_dim: int, num_attention_heads: int,
                 depthwise_conv_kernel_size: int, dropout: float = 0.0):
        super().__init__()
        self.ffn1 = _FeedForwardModule(input_dim, ffn_dim, dropout=dropout)
        self.self_attn_layer_norm = nn.LayerNorm(input_dim)
        self.self_attn = nn.MultiheadAttention(input_dim, num_attention_heads, dropout=dropout)
        self.self_attn_dropout = nn.Dropout(dropout)
        self.conv_module = _ConvolutionModule(
            input_dim=input_dim, num_channels=input_dim,
            depthwise_kernel_size=depthwise_conv_kernel_size, dropout=dropout, bias=True)
        self.ffn2 = _FeedForwardModule(input_dim, ffn_dim, dropout=dropout)
        self.final_layer_norm = nn.LayerNorm(input_dim)

    def forward(self, x: Tensor, key_padding_mask: Optional[Tensor]) -> Tensor:  # x: [T, B, D]
        x = 0.5 * self.ffn1(x) + x
        a = self.self_attn_layer_norm(x)
        a, _ = self.self_attn(query=a, key=a, value=a, key_padding_mask=key_padding_mask,
                              need_weights=False)
        x = self.self_attn_dropout(a) + x
        x = x + self.conv_module(x.transpose(0, 1)).transpose(0, 1)
        x = 0.5 * self.ffn2(x) + x
        return self.final_layer_norm(x)


class Conformer(nn.Module):
    """torchaudio.models.Conformer signature: forward(input[B,T,D], lengths[B]) -> (out, lengths)."""

    def __init__(self, input_dim: int, num_heads: int, ffn_dim: int, num_layers: int,
                 depthwise_conv_kernel_size: int, dropout: float = 0.0,
                 use_group_norm: bool = False, convolution_first: bool = False):
        super().__init__()
        if use_group_norm or convolution_first:
            raise NotImplementedError("reference call sites use the defaults (early_exit.py:603-615)")
        self.conformer_layers = nn.ModuleList(
            [ConformerLayer(input_dim, ffn_dim, num_heads, depthwise_conv_kernel_size, dropout=dropout)
             for _ in range(num_layers)])

    def forward(self, input: Tensor, lengths: Tensor) -> Tuple[Tensor, Tensor]:
        mask = lengths_to_padding_mask(lengths)
        x = input.transpose(0, 1)
        for layer in self.conformer_layers:
            x = layer(x, mask)
        return x.transpose(0, 1), lengths


# --------------------------------------------------------------------------
# reference-owned pieces
# --------------------------------------------------------------------------
class Subsample(nn.Module):
    """Two Conv1d(k=3, s=2, p=0), no activation between (early_exit.py:24-48)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.sequential = nn.Sequential(
            nn.Conv1d(in_channels, out_channels, kernel_size=3, stride=2, padding=0),
            nn.Conv1d(out_channels, out_channels, kernel_size=3, stride=2, padding=0),
        )

    def forward(self, x: Tensor) -> Tensor:
        return self.sequential(x)


def sinusoid_table(max_len: int, d_model: int) -> Tensor:
    """pe[t,0,2i]=sin(t*w_i), pe[t,0,2i+1]=cos(t*w_i), w_i=exp(-2i*ln(1e4)/D)
    (positional_encoding.py:59-64).  Built with the same torch ops/order so the
    fp32 table is bit-identical."""
    pos = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, 1, d_model)
    pe[:, 0, 0::2] = torch.sin(pos * div)
    pe[:, 0, 1::2] = torch.cos(pos * div)
    return pe


class SinusoidPE(nn.Module):
    """x[b,t,:] += pe[t]; dropout (positional_encoding.py:65-73).  ``pe`` is a buffer."""

    def __init__(self, d_model: int, dropout: float, max_len: int):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        self.register_buffer("pe", sinusoid_table(max_len, d_model))

    def forward(self, x: Tensor) -> Tensor:  # [B, T, D]
        return self.dropout(x + self.pe[: x.size(1), 0].unsqueeze(0))


def encoder_lengths(lengths: Tensor, t_out: int) -> Tensor:
    """clamp(lengths / 4, max=T').to(int): true division then truncation (early_exit.py:623)."""
    return torch.clamp(lengths / 4, max=t_out).to(torch.int)


class EarlyConformerRef(nn.Module):
    """Restatement of Early_conformer (early_exit.py:565-634), same ctor kwargs."""

    def __init__(self, src_pad_idx, n_enc_exits, enc_voc_size, dec_voc_size, d_model, n_head,
                 max_len, d_feed_forward, n_enc_layers, features_length, drop_prob,
                 depthwise_kernel_size, device="cpu"):
        super().__init__()
        self.n_enc_exits = n_enc_exits
        self.conv_subsample = Subsample(features_length, d_model)
        self.positional_encoder = SinusoidPE(d_model, drop_prob, max_len)
        self.linears = nn.ModuleList([nn.Linear(d_model, dec_voc_size) for _ in range(n_enc_exits)])
        self.conformer = nn.ModuleList([
            Conformer(input_dim=d_model, num_heads=n_head, ffn_dim=d_feed_forward,
                      num_layers=n_enc_layers, depthwise_conv_kernel_size=depthwise_kernel_size,
                      dropout=drop_prob)
            for _ in range(n_enc_exits)])

    def stem(self, src: Tensor) -> Tensor:
        return self.positional_encoder(self.conv_subsample(src).permute(0, 2, 1))

    def forward(self, src: Tensor, lengths: Tensor, return_taps: bool = False):
        enc = self.stem(src)
        length = encoder_lengths(lengths, enc.size(1))
        outs: List[Tensor] = []
        taps: List[Tensor] = []
        for head, group in zip(self.linears, self.conformer):
            enc, _ = group(enc, length)
            taps.append(enc)
            outs.append(F.log_softmax(head(enc), dim=2).unsqueeze(0))
        out = torch.cat(outs)
        return (out, torch.stack(taps)) if return_taps else out


class SplitformerRef(EarlyConformerRef):
    """Restatement of Splitformer (early_exit.py:227-364): Early_conformer plus, at the FIRST and the LAST exit, a
    one-layer Conformer that runs on the 2x time-down-sampled input of that exit group and is added back
    (nearest-neighbour up-sampled) to the group's output before the head.  Reference quirks kept on purpose:
    the branch's key lengths come from the MEL lengths, ``clamp((lengths + pad) / 2, max=T'/2)`` (:324-331), not
    from the encoder lengths; ``index // (n_enc_exits - 1)`` picks the branch (:320), so n_enc_exits >= 2."""

    factor = 2

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        first = self.conformer[0].conformer_layers[0]
        d_model, n_head = first.self_attn.embed_dim, first.self_attn.num_heads
        d_ff = first.ffn1.sequential[1].out_features
        dw = first.conv_module.sequential[2].kernel_size[0]
        p = first.ffn1.sequential[3].p
        self.conformer_parallel = nn.ModuleList([
            Conformer(input_dim=d_model, num_heads=n_head, ffn_dim=d_ff, num_layers=1,
                      depthwise_conv_kernel_size=dw, dropout=p) for _ in range(2)])

    def forward(self, src: Tensor, lengths: Tensor) -> Tensor:  # type: ignore[override]
        enc = self.stem(src)
        base = encoder_lengths(lengths, enc.size(1))
        outs: List[Tensor] = []
        last = self.n_enc_exits - 1
        for index, (head, group) in enumerate(zip(self.linears, self.conformer)):
            side = enc  # the group's INPUT feeds the parallel branch
            enc, _ = group(enc, base)
            if index in (0, last):
                pad = (-side.size(1)) % self.factor
                if pad:
                    side = torch.cat((side, side.new_zeros(side.size(0), pad, side.size(2))), dim=1)
                side = side[:, :: self.factor, :]
                side_len = torch.clamp((lengths + pad) / self.factor, max=side.size(1)).to(torch.int)
                side, _ = self.conformer_parallel[index // last](side, side_len)
                side = torch.repeat_interleave(side, self.factor, dim=1)
                if pad:
                    side = side[:, :-pad, :]
                enc = enc + side
            outs.append(F.log_softmax(head(enc), dim=2).unsqueeze(0))
        return torch.cat(outs)


class _SingleConvStem(nn.Module):
    """Conv1dSubampling_Zipformer (early_exit.py:80-95): one Conv1d(k=3, s=2)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=3, stride=2, padding=0)

    def forward(self, x: Tensor) -> Tensor:
        return self.conv(x)


class EarlyZipformerRef(nn.Module):
    """Restatement of Early_zipformer (early_exit.py:117-224): one-convolution stem (T1 = (T-3)//2+1 frames), two
    Conformer groups at full rate, then five stacks of groups at 1/2, 1/4, 1/8, 1/4, 1/2 of the frame rate -- each
    stack: zero-pad to a multiple of the factor, keep every factor-th frame, run its groups, repeat every frame
    factor times, drop the padding, add the stack's input -- and ONE head on every second frame.  The module list
    has n_enc_exits groups of n_enc_layers layers and the forward indexes groups 0 .. 18, so n_enc_exits >= 19.
    Reference quirk kept: inside the stacks the key lengths are ``clamp((mel_lengths + pad) / factor, max=T_stack)``."""

    factors = (2, 4, 8, 4, 2)
    stack = (2, 4, 5, 4, 2)

    def __init__(self, src_pad_idx, n_enc_exits, enc_voc_size, dec_voc_size, d_model, n_head,
                 max_len, d_feed_forward, n_enc_layers, features_length, drop_prob,
                 depthwise_kernel_size, device="cpu"):
        super().__init__()
        self.conv_subsample = _SingleConvStem(features_length, d_model)
        self.positional_encoder = SinusoidPE(d_model, drop_prob, max_len)
        self.linear = nn.Linear(d_model, dec_voc_size)
        self.conformer = nn.ModuleList([
            Conformer(input_dim=d_model, num_heads=n_head, ffn_dim=d_feed_forward,
                      num_layers=n_enc_layers, depthwise_conv_kernel_size=depthwise_kernel_size,
                      dropout=drop_prob)
            for _ in range(n_enc_exits)])

    def forward(self, src: Tensor, lengths: Tensor) -> Tensor:
        enc = self.positional_encoder(self.conv_subsample(src).permute(0, 2, 1))
        base = torch.clamp(lengths / 2, max=enc.size(1)).to(torch.int)
        enc, _ = self.conformer[0](enc, base)
        enc, _ = self.conformer[1](enc, base)
        first = 2
        for factor, count in zip(self.factors, self.stack):
            skip = enc
            pad = (-enc.size(1)) % factor
            if pad:
                enc = torch.cat((enc, enc.new_zeros(enc.size(0), pad, enc.size(2))), dim=1)
            enc = enc[:, ::factor, :]
            length = torch.clamp((lengths + pad) / factor, max=enc.size(1)).to(torch.int)
            for group in self.conformer[first:first + count]:
                enc, _ = group(enc, length)
            first += count
            enc = torch.repeat_interleave(enc, factor, dim=1)
            if pad:
                enc = enc[:, :-pad, :]
            enc = enc + skip
        return F.log_softmax(self.linear(enc[:, ::2, :]), dim=2).unsqueeze(0)


def trace_substeps(model: "EarlyConformerRef", src: Tensor, lengths: Tensor) -> List[Tensor]:
    """Residual stream [B, T', D] after the stem and after every sub-step of every layer
    (ffn1, attention, conv, ffn2+final LN) -- the checkpoints ``eec_encoder_forward(stop_after=k)``
    exposes, so a GPU mismatch can be localised to one kernel group."""
    enc = model.stem(src)
    out = [enc]
    mask = lengths_to_padding_mask(encoder_lengths(lengths, enc.size(1)))
    x = enc.transpose(0, 1)
    for group in model.conformer:
        for layer in group.conformer_layers:
            x = 0.5 * layer.ffn1(x) + x
            out.append(x.transpose(0, 1))
            a = layer.self_attn_layer_norm(x)
            a, _ = layer.self_attn(a, a, a, key_padding_mask=mask, need_weights=False)
            x = a + x
            out.append(x.transpose(0, 1))
            x = x + layer.conv_module(x.transpose(0, 1)).transpose(0, 1)
            out.append(x.transpose(0, 1))
            x = layer.final_layer_norm(0.5 * layer.ffn2(x) + x)
            out.append(x.transpose(0, 1))
    return out


class FullConformerEncoderRef(nn.Module):
    """Encoder half of full_conformer (early_exit.py:637-737, 764-800): same stem
    and Conformer groups, heads named ``linears_1``, PE named ``positional_encoder_1``.
    The AED decoder (``emb``, ``decoders``, ``linears_2``...) is out of scope."""

    def __init__(self, n_enc_exits, dec_voc_size, d_model, n_head, max_len, d_feed_forward,
                 n_enc_layers, features_length, drop_prob, depthwise_kernel_size):
        super().__init__()
        self.conv_subsample = Subsample(features_length, d_model)
        self.linears_1 = nn.ModuleList([nn.Linear(d_model, dec_voc_size) for _ in range(n_enc_exits)])
        self.positional_encoder_1 = SinusoidPE(d_model, drop_prob, max_len)
        self.conformer = nn.ModuleList([
            Conformer(input_dim=d_model, num_heads=n_head, ffn_dim=d_feed_forward,
                      num_layers=n_enc_layers, depthwise_conv_kernel_size=depthwise_kernel_size,
                      dropout=drop_prob)
            for _ in range(n_enc_exits)])

    def _encoder_(self, src: Tensor, lengths: Tensor, layer_n: int) -> Tensor:
        enc = self.positional_encoder_1(self.conv_subsample(src).permute(0, 2, 1))
        length = encoder_lengths(lengths, enc.size(1))
        for i, group in enumerate(self.conformer, start=1):
            enc, _ = group(enc, length)
            if i == layer_n:
                break
        return enc

    def encoder_logprobs(self, src: Tensor, lengths: Tensor) -> Tensor:
        enc = self.positional_encoder_1(self.conv_subsample(src).permute(0, 2, 1))
        length = encoder_lengths(lengths, enc.size(1))
        outs = []
        for head, group in zip(self.linears_1, self.conformer):
            enc, _ = group(enc, length)
            outs.append(F.log_softmax(head(enc), dim=2).unsqueeze(0))
        return torch.cat(outs)


# --------------------------------------------------------------------------
# callers' arithmetic on the path
# --------------------------------------------------------------------------
def greedy_ctc(emission: Tensor, blank: int = 0) -> List[int]:
    """argmax -> unique_consecutive -> drop blank, un-batched [T', V] (beam_infer.py:9-24).
    Ignores lengths, exactly like the reference."""
    idx = torch.unique_consecutive(torch.argmax(emission, dim=-1), dim=-1)
    return [int(i) for i in idx if int(i) != blank]


def summed_exit_ctc_loss(enc_out: Tensor, targets: Tensor, target_len: Tensor) -> Tensor:
    """sum_e CTCLoss(blank=0, 'mean', zero_infinity=True)(enc_out[e].permute(1,0,2), targets,
    input_len = T' for every b, target_len)   (train.py:53-65, ctor train.py:259)."""
    ctc = nn.CTCLoss(blank=0, reduction="mean", zero_infinity=True)
    b, t = enc_out.size(1), enc_out.size(2)
    in_len = torch.full((b,), t, dtype=torch.long)
    loss = enc_out.new_zeros(())
    for e in range(enc_out.size(0)):
        loss = loss + ctc(enc_out[e].permute(1, 0, 2), targets, in_len, target_len)
    return loss


def xavier_like_reference(model: nn.Module) -> None:
    """``model.apply(initialize_weights)`` semantics (util/model_utils.py:10-12):
    xavier_uniform_ on every sub-module ``.weight`` with dim > 1."""
    def _init(m):
        if hasattr(m, "weight") and isinstance(m.weight, Tensor) and m.weight.dim() > 1:
            nn.init.xavier_uniform_(m.weight.data)
    model.apply(_init)
