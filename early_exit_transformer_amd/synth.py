"""Portable synthetic weights / mel batches (counter-based, independent of torch's RNG).

Every value is a pure function of (seed, tensor name, flat index) through
SplitMix64, so the build container, the GPU box and any later round regenerate
bit-identical fp32 tensors.  Distributions follow the reference's start-up
state: ``model.apply(initialize_weights)`` = xavier-uniform on every >=2-D
``.weight`` (/root/reference/util/model_utils.py:10-12, train.py:230), torch
defaults elsewhere.  ``style="trained"`` additionally perturbs LayerNorm /
BatchNorm affine+statistics and the attention biases, which a fresh init leaves
at 1/0 -- parity tests use it so that BN folding, LN affine and bias paths are
actually exercised.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Mapping

import numpy as np
import torch

_U64 = np.uint64


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def _key(seed: int, name: str) -> np.uint64:
    h = zlib.crc32(name.encode()) | (zlib.adler32(name.encode()) << 32)
    with np.errstate(over="ignore"):
        return _splitmix64(np.array([(h ^ (seed * 0x51ED2705)) & 0xFFFFFFFFFFFFFFFF], dtype=_U64))[0]


def uniform01(seed: int, name: str, n: int, stream: int = 0) -> np.ndarray:
    """n float64 in [0,1), reproducible for (seed, name, stream)."""
    with np.errstate(over="ignore"):
        base = _key(seed, name) + _U64((stream * 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF)
        bits = _splitmix64(base + np.arange(n, dtype=_U64))
    return (bits >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal(seed: int, name: str, n: int) -> np.ndarray:
    u1 = uniform01(seed, name, n, stream=1)
    u2 = uniform01(seed, name, n, stream=2)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)


def _t(a: np.ndarray, shape) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).reshape(tuple(shape))


def synth_state_dict(template: Mapping[str, torch.Tensor], seed: int = 0, style: str = "init",
                     head_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """New tensors for every entry of ``template`` (a model.state_dict()).

    ``head_scale`` multiplies the per-exit head weights (``linears*.weight``) so
    log-probs are peaky enough for exact greedy-decode comparisons.
    """
    if style not in ("init", "trained"):
        raise ValueError(style)
    trained = style == "trained"
    out: Dict[str, torch.Tensor] = {}
    for name, ref in template.items():
        shape, n = tuple(ref.shape), ref.numel()
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "pe":
            out[name] = ref.clone()
        elif leaf == "num_batches_tracked":
            out[name] = torch.zeros_like(ref)
        elif leaf == "running_mean":
            out[name] = _t(0.1 * normal(seed, name, n) if trained else np.zeros(n), shape)
        elif leaf == "running_var":
            out[name] = _t(0.5 + uniform01(seed, name, n) if trained else np.ones(n), shape)
        elif leaf in ("weight", "in_proj_weight") and ref.dim() >= 2:
            rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
            bound = math.sqrt(6.0 / (shape[1] * rf + shape[0] * rf))
            w = (2.0 * uniform01(seed, name, n) - 1.0) * bound
            if head_scale != 1.0 and name.split(".")[0] in ("linears", "linears_1"):
                w = w * head_scale
            out[name] = _t(w, shape)
        elif leaf == "weight":  # LayerNorm / BatchNorm gamma
            out[name] = _t(1.0 + 0.1 * normal(seed, name, n) if trained else np.ones(n), shape)
        elif leaf in ("bias", "in_proj_bias"):
            wname = name[: -len(leaf)] + "weight"
            w = template.get(wname)
            mha_bias = leaf == "in_proj_bias" or name.endswith("out_proj.bias")
            if mha_bias:
                out[name] = _t(0.05 * normal(seed, name, n) if trained else np.zeros(n), shape)
            elif w is not None and w.dim() >= 2:  # Linear / Conv default U(+-1/sqrt(fan_in))
                fan_in = int(np.prod(w.shape[1:]))
                out[name] = _t((2.0 * uniform01(seed, name, n) - 1.0) / math.sqrt(fan_in), shape)
            else:  # LayerNorm / BatchNorm beta
                out[name] = _t(0.1 * normal(seed, name, n) if trained else np.zeros(n), shape)
        else:
            raise KeyError(f"no synthetic rule for state_dict entry {name!r}")
        out[name] = out[name].to(ref.dtype)
    return out


def synth_mel(batch: int, n_mels: int, frames: int, seed: int = 0, kind: str = "lognormal") -> torch.Tensor:
    """fp32 [B, n_mels, T].  ``lognormal``: exp(N(-2, 2)) clipped to [0, 1e4] -- the
    dynamic range of the reference's un-logged power mel (util/data_loader.py:7-18)."""
    n = batch * n_mels * frames
    if kind == "lognormal":
        x = np.clip(np.exp(-2.0 + 2.0 * normal(seed, "mel", n)), 0.0, 1e4)
    elif kind == "uniform":
        x = uniform01(seed, "mel", n)
    else:
        raise ValueError(kind)
    return _t(x, (batch, n_mels, frames))


def synth_lengths(batch: int, frames: int, seed: int = 0, ragged: bool = True) -> torch.Tensor:
    """int64 [B] valid mel lengths, sorted descending, max == frames (the reference's
    collate pads to the longest utterance, so max(lengths) == T always holds)."""
    if not ragged or batch == 1:
        return torch.full((batch,), frames, dtype=torch.int64)
    u = uniform01(seed, "lengths", batch)
    lens = np.floor(frames * (0.5 + 0.5 * u)).astype(np.int64)
    lens[0] = frames
    return torch.from_numpy(np.sort(lens)[::-1].copy())


def synth_targets(batch: int, max_len: int, vocab: int, seed: int = 0):
    """CTC targets as the reference collate builds them (util/data_loader.py:207-225):
    BOS(1) ids... EOS(2), PAD(126)-padded 2-D int64, plus target lengths incl. BOS/EOS."""
    u = uniform01(seed, "tgt_len", batch)
    lens = np.maximum(3, np.floor(max_len * (0.5 + 0.5 * u))).astype(np.int64)
    lens[0] = max_len
    ids = np.full((batch, max_len), 126, dtype=np.int64)
    pool = np.array([i for i in range(3, vocab) if i not in (126, 127)], dtype=np.int64)
    for b in range(batch):
        k = int(lens[b])
        r = uniform01(seed, f"tgt{b}", k)
        ids[b, :k] = pool[(r * len(pool)).astype(np.int64)]
        ids[b, 0], ids[b, k - 1] = 1, 2
    return torch.from_numpy(ids), torch.from_numpy(lens)
