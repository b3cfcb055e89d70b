"""Parameter containers with torchaudio's ``Conformer`` module tree.

The reference builds its encoder groups from ``torchaudio.models.conformer.Conformer``
(/root/reference/models/model/early_exit.py:16,603-615).  Checkpoints written by the
reference therefore carry torchaudio's parameter names; these classes reproduce that tree
(attribute names and ``Sequential`` indices) so such checkpoints load
with ``strict=True`` and ``initialize_weights`` / optimizers see the same parameters.

They hold parameters only.  The arithmetic of a group lives in the HIP library and is
driven by the owning model (``model.Early_conformer``) for the whole stack at once; calling
a container's ``forward`` directly raises -- there is no PyTorch fallback path.
"""
from __future__ import annotations

from torch import nn


class _HipOnly(nn.Module):
    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container: its arithmetic runs inside libeec.so and is "
            "launched by Early_conformer/full_conformer.forward for the whole encoder stack.")


class _FeedForwardModule(_HipOnly):
    def __init__(self, input_dim: int, hidden_dim: int, dropout: float = 0.0):
        super().__init__()
        self.sequential = nn.Sequential(
            nn.LayerNorm(input_dim), nn.Linear(input_dim, hidden_dim, bias=True), nn.SiLU(), nn.Dropout(dropout),
            nn.Linear(hidden_dim, input_dim, bias=True), nn.Dropout(dropout))


class _ConvolutionModule(_HipOnly):
    def __init__(self, input_dim: int, num_channels: int, depthwise_kernel_size: int, dropout: float = 0.0):
        super().__init__()
        if depthwise_kernel_size % 2 != 1:
            raise ValueError("depthwise_kernel_size must be odd to achieve 'SAME' padding.")
        self.layer_norm = nn.LayerNorm(input_dim)
        self.sequential = nn.Sequential(
            nn.Conv1d(input_dim, 2 * num_channels, 1, bias=True), nn.GLU(dim=1),
            nn.Conv1d(num_channels, num_channels, depthwise_kernel_size, padding=(depthwise_kernel_size - 1) // 2,
                      groups=num_channels, bias=True),
            nn.BatchNorm1d(num_channels), nn.SiLU(), nn.Conv1d(num_channels, input_dim, 1, bias=True),
            nn.Dropout(dropout))


class ConformerLayer(_HipOnly):
    def __init__(self, input_dim: int, ffn_dim: int, num_attention_heads: int, depthwise_conv_kernel_size: int,
                 dropout: float = 0.0):
        super().__init__()
        self.ffn1 = _FeedForwardModule(input_dim, ffn_dim, dropout)
        self.self_attn_layer_norm = nn.LayerNorm(input_dim)
        self.self_attn = nn.MultiheadAttention(input_dim, num_attention_heads, dropout=dropout)
        self.self_attn_dropout = nn.Dropout(dropout)
        self.conv_module = _ConvolutionModule(input_dim, input_dim, depthwise_conv_kernel_size, dropout)
        self.ffn2 = _FeedForwardModule(input_dim, ffn_dim, dropout)
        self.final_layer_norm = nn.LayerNorm(input_dim)


class Conformer(_HipOnly):
    """Same constructor keywords as torchaudio.models.Conformer at the reference's call sites."""

    def __init__(self, input_dim: int, num_heads: int, ffn_dim: int, num_layers: int,
                 depthwise_conv_kernel_size: int, dropout: float = 0.0):
        super().__init__()
        self.conformer_layers = nn.ModuleList(
            [ConformerLayer(input_dim, ffn_dim, num_heads, depthwise_conv_kernel_size, dropout)
             for _ in range(num_layers)])
