"""Mel front end on the device: drop-in for the reference's ``spec_transform`` + ``melspec_transform``
(util/data_loader.py:7-18: torchaudio Spectrogram(n_fft=1024, hop 160, win 320) -> MelScale(80 bins), un-logged power).
The arithmetic is one HIP kernel (csrc/frontend.hip, exact-fp32 MFMA DFT + mel filters); there is no CPU path."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from . import capi


class MelFrontend:
    """``MelFrontend(args)(wave[B, L], lengths[B] | None) -> mel [B, n_mels, 1 + L // hop]``; ``args`` carries the
    reference's flags ``sample_rate, n_fft, win_length, hop_length, n_mels`` (util/conf.py:334-381; note the reference
    doubles ``n_fft`` for the transform).  Keyword arguments override / replace ``args``."""

    def __init__(self, args=None, sample_rate: int = 16000, n_fft: int = 512, win_length: int = 320, hop_length: int = 160,
                 n_mels: int = 80):
        g = lambda name, default: getattr(args, name, default) if args is not None else default  # noqa: E731
        self.sample_rate, self.n_fft = g("sample_rate", sample_rate), g("n_fft", n_fft)
        self.win_length, self.hop_length, self.n_mels = g("win_length", win_length), g("hop_length", hop_length), g("n_mels", n_mels)
        self._fe = None
        self._device = None

    def __del__(self):
        if getattr(self, "_fe", None) is not None:
            try:
                capi.load().eec_frontend_destroy(self._fe)
            except Exception:
                pass

    def _handle(self, device: torch.device):
        lib = capi.load()
        if self._fe is not None and self._device != device:
            lib.eec_frontend_destroy(self._fe)
            self._fe = None
        if self._fe is None:
            h = C.c_void_p()
            with torch.cuda.device(device):
                rc = lib.eec_frontend_create(self.sample_rate, 2 * self.n_fft, self.win_length, self.hop_length, self.n_mels, C.byref(h))
            if rc != 0:
                raise RuntimeError(f"eec_frontend_create failed (code {rc}): {lib.eec_frontend_last_error().decode()}")
            self._fe, self._device = h, device
        return self._fe

    def frames(self, n_samples: int) -> int:
        return 1 + n_samples // self.hop_length if n_samples > 0 else 0

    def __call__(self, wave: Tensor, lengths: Optional[Tensor] = None) -> Tensor:
        if not wave.is_cuda:
            raise RuntimeError("the mel front end runs on a HIP device only (the CPU reference lives in oracle/)")
        squeeze = wave.dim() == 1
        if squeeze:
            wave = wave.unsqueeze(0)
        wave = wave.contiguous().float()
        B, L = wave.shape
        dev = wave.device
        mel = torch.empty((B, self.n_mels, self.frames(L)), dtype=torch.float32, device=dev)
        len_dev = lengths.to(device=dev, dtype=torch.int64).contiguous() if lengths is not None else None
        with torch.cuda.device(dev):
            fe = self._handle(dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            lib = capi.load()
            rc = lib.eec_frontend_forward(fe, wave.data_ptr(), len_dev.data_ptr() if len_dev is not None else None, B, L,
                                          mel.data_ptr(), C.c_void_p(stream))
            if rc != 0:
                raise RuntimeError(f"eec_frontend_forward failed (code {rc}): {lib.eec_frontend_last_error().decode()}")
            wave.record_stream(torch.cuda.current_stream(dev))
        return mel[0] if squeeze else mel
