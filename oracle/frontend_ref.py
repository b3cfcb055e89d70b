"""fp32 torch-CPU restatement of the reference's mel front end (the oracle for csrc/frontend.hip).

TEST INFRASTRUCTURE -- see oracle/__init__.py for who may import this.

Reference: util/data_loader.py:7-18
    spec = torchaudio.transforms.Spectrogram(n_fft=args.n_fft * 2, hop_length=args.hop_length, win_length=args.win_length)(wave)
    mel  = torchaudio.transforms.MelScale(sample_rate=args.sample_rate, n_mels=args.n_mels, n_stft=args.n_fft + 1)(spec)
with util/conf.py defaults n_fft 512 (-> a 1024-point transform, 513 bins), win_length 320, hop_length 160, 80 mel bins,
16 kHz.  torchaudio is a third-party dependency that is NOT in the reference tree and not installed here (version
unpinned): both transforms are restated from their published definitions --

* ``Spectrogram`` defaults: hann window (periodic), power 2, not normalised, center=True with reflect padding, one-sided:
  ``|torch.stft(wave, n_fft, hop, win_length, window=hann(win_length), center=True, pad_mode="reflect")| ** 2``.
* ``MelScale`` defaults: f_min 0, f_max sample_rate // 2, norm None, mel_scale "htk":
  ``melscale_fbanks``: triangular filters between points equally spaced on m = 2595 log10(1 + f / 700).

Parity unpinned by the reference (it holds no vectors for the front end); torch.stft is the installed torch's kernel.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> Tensor:
    """[n_freqs, n_mels] triangular filterbank, htk mel scale, no area normalisation."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def mel_frontend(wave: Tensor, sample_rate: int = 16000, n_fft: int = 512, win_length: int = 320, hop_length: int = 160,
                 n_mels: int = 80) -> Tensor:
    """wave [L] or [B, L] fp32 -> power mel [n_mels, T] / [B, n_mels, T], T = 1 + L // hop (un-logged, as the reference)."""
    nfft = 2 * n_fft
    spec = torch.stft(wave, nfft, hop_length, win_length, window=torch.hann_window(win_length), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True).abs().pow(2.0)
    fb = melscale_fbanks(nfft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate)
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)


def mel_frontend_batch(wave: Tensor, lengths: Tensor, **kw) -> Tensor:
    """Per-utterance front end + zero padding to the longest, as the reference's collate does (data_loader.py:20-26,
    pad_sequence with 0): wave [B, Lmax], lengths [B] -> [B, n_mels, 1 + Lmax // hop]."""
    hop = kw.get("hop_length", 160)
    outs = [mel_frontend(wave[b, : int(lengths[b])], **kw) for b in range(wave.size(0))]
    T = max(o.size(1) for o in outs)
    out = torch.zeros(wave.size(0), outs[0].size(0), T)
    for b, o in enumerate(outs):
        out[b, :, : o.size(1)] = o
    return out
