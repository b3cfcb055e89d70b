"""Drop-in mirrors of the reference's ``Early_conformer`` / ``full_conformer``.

Same constructor keywords, ``forward`` signatures, return shapes and state_dict keys as
/root/reference/models/model/early_exit.py:565-634 (Early_conformer) and :637-811
(full_conformer), as called from train.py:148-178,37,54 and inference.py:45-46,66.
The encoder stack (subsampling, positional encoding, length mask, E x L Conformer layers,
per-exit Linear + log_softmax) is ONE call into libeec.so on the caller's current HIP
stream; there is no PyTorch implementation of it in this package and no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import capi
from .conformer import Conformer


class PositionalEncoding(nn.Module):
    """Holder of the sinusoid table buffer ``pe`` [max_len, 1, d_model] (reference
    models/embedding/positional_encoding.py:55-64).  The add happens inside the stem kernel."""

    def __init__(self, d_model: int, dropout: float, max_len: int):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        t = torch.arange(max_len).unsqueeze(1)
        w = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(t * w)
        pe[:, 0, 1::2] = torch.cos(t * w)
        self.register_buffer("pe", pe)

    def forward(self, x: Tensor) -> Tensor:  # used by the AED decoder side only: [B, S, D]
        return self.dropout(x + self.pe[: x.size(1), 0].unsqueeze(0))


class Conv1dSubampling(nn.Module):
    """Parameter holder for the two Conv1d(k=3, s=2) of the stem (early_exit.py:24-48)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.sequential = nn.Sequential(
            nn.Conv1d(in_channels, out_channels, kernel_size=3, stride=2, padding=0),
            nn.Conv1d(out_channels, out_channels, kernel_size=3, stride=2, padding=0))


def _layer_params(ptr, prefix: str, n_groups: int, n_layers: int):
    """HOST array of EecLayerParams for ``{prefix}.{g}.conformer_layers.{l}.*`` (group-major)."""
    layers = (capi.EecLayerParams * (n_groups * n_layers))()
    for g in range(n_groups):
        for l in range(n_layers):
            lp = layers[g * n_layers + l]
            for field, suffix in capi.LAYER_KEYS.items():
                setattr(lp, field, ptr(f"{prefix}.{g}.conformer_layers.{l}.{suffix}"))
    return layers


def _to_device(t: Tensor, dev: torch.device, dtype: torch.dtype = torch.int64) -> Tensor:
    """``t.to(dev, dtype).contiguous()``; a small CPU int64 tensor (the collate's ``lengths`` are CPU tensors in the
    reference, train.py:34,54) travels in a kernel's argument block instead (eec_upload_i64): a host-to-device copy in front
    of the forward drains the host's launch queue and leaves a hole on the stream once per step."""
    if t.is_cuda or dev.type != "cuda" or dtype != torch.int64 or t.numel() == 0:
        return t.to(device=dev, dtype=dtype).contiguous()
    lib = capi.load()
    if t.numel() > lib.eec_upload_i64_max():
        return t.to(device=dev, dtype=dtype).contiguous()
    dev = torch.device("cuda", torch.cuda.current_device()) if dev.index is None else dev
    host = t.to(torch.int64).contiguous()
    out = torch.empty(host.shape, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        capi.check(lib.eec_upload_i64(host.data_ptr(), host.numel(), out.data_ptr(),
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "eec_upload_i64")
    return out


class _HipEncoderMixin:
    """Owns the libeec encoder handle, the packed-weight cache and the workspace."""

    _head_attr = "linears"
    _pe_attr = "positional_encoder"
    # default operand mode: the one that keeps the north-star tolerance (|d log-prob| <= 1e-3, FLAT) on every committed fixture,
    # the peaky (trained-like) one included.  "f16f8" is the faster opt-in: 1e-3 on near-uniform outputs only
    precision = "f16x3"
    train_passes = 3     # training GEMMs: 3 = bf16 hi/lo split, three MFMA products (~fp32 results); 1 = plain bf16 operands

    def _hip_init(self, d_model, n_head, d_ff, dw_kernel, n_exits, n_layers, n_mels, vocab, max_len):
        self._cfg = capi.EecConfig(d_model, n_head, d_ff, dw_kernel, n_exits, n_layers, n_mels, vocab, max_len,
                                   capi.ARCH_CONFORMER)
        self._enc = None
        self._enc_device = None
        self._packed_key = None
        self._ws: Dict[Tuple[int, int, int], Tensor] = {}
        self._gws: Dict[Tuple[int, int, int], Tensor] = {}  # workspaces of the group-level entry points
        self._keep: list = []

    def __del__(self):
        enc = getattr(self, "_enc", None)
        if enc is not None:
            try:
                capi.load().eec_encoder_destroy(enc)
            except Exception:
                pass
        tr = getattr(self, "_trainer", None)
        if tr is not None:
            try:
                capi.load().eec_trainer_destroy(tr)
            except Exception:
                pass

    # -- packing ------------------------------------------------------------
    def _param_tensors(self) -> List[Tensor]:
        return list(self.conv_subsample.parameters()) + list(self.conformer.parameters()) + \
            list(self.conformer.buffers()) + list(getattr(self, self._head_attr).parameters()) + \
            [getattr(self, self._pe_attr).pe]

    def _ensure_packed(self, device: torch.device) -> None:
        tensors = self._param_tensors()
        key = (device, tuple(t._version for t in tensors), tuple(t.data_ptr() for t in tensors))
        if self._enc is not None and key == self._packed_key:
            return
        lib = capi.load()
        if self._enc is not None and self._enc_device != device:
            # model.to(another device): the packed-weight arena lives on the old device (one handle per device, eec.h)
            lib.eec_encoder_destroy(self._enc)
            self._enc, self._packed_key = None, None
            self._ws.clear()
            self._gws.clear()
        if self._enc is None:
            h = C.c_void_p()
            capi.check(lib.eec_encoder_create(C.byref(self._cfg), C.byref(h)), "eec_encoder_create")
            self._enc, self._enc_device = h, device
        sd = {k: v for k, v in self.state_dict(keep_vars=True).items()}

        def ptr(name: str) -> int:
            t = sd[name]
            if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError(f"parameter {name} must be a contiguous fp32 tensor on {device}")
            return t.data_ptr()

        self._pack(lib, device, sd, ptr)
        self._packed_key = key

    def _pack(self, lib, device, sd, ptr) -> None:
        E, L = self._cfg.n_exits, self._cfg.layers_per_exit
        layers = _layer_params(ptr, "conformer", E, L)
        hw = (C.c_void_p * E)(*[ptr(f"{self._head_attr}.{e}.weight") for e in range(E)])
        hb = (C.c_void_p * E)(*[ptr(f"{self._head_attr}.{e}.bias") for e in range(E)])
        params = capi.EecParams(ptr("conv_subsample.sequential.0.weight"), ptr("conv_subsample.sequential.0.bias"),
                                ptr("conv_subsample.sequential.1.weight"), ptr("conv_subsample.sequential.1.bias"),
                                ptr(f"{self._pe_attr}.pe"), layers, hw, hb)
        stream = torch.cuda.current_stream(device).cuda_stream
        capi.check(lib.eec_encoder_pack(self._enc, C.byref(params), C.c_void_p(stream)), "eec_encoder_pack")

    def _group(self, enc_handle, group: int, x: Tensor, key_len: Tensor) -> None:
        """x [B, T', D] fp32 contiguous, in place; key_len [B] int32 on the device."""
        lib, dev = capi.load(), x.device
        B, Tq, _ = x.shape
        k = (B, Tq, dev.index or 0)
        ws = self._gws.get(k)
        if ws is None:
            if len(self._gws) > 4:
                self._gws.clear()
            ws = torch.empty(lib.eec_encoder_group_workspace_bytes(enc_handle, B, Tq) + 256, dtype=torch.uint8, device=dev)
            self._gws[k] = ws
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.eec_encoder_group_forward(enc_handle, group, x.data_ptr(), key_len.data_ptr(), B, Tq,
                                           capi.PRECISIONS[self.precision], ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()),
                                           C.c_void_p(stream))
        capi.check(rc, "eec_encoder_group_forward")

    # -- measurement hook -----------------------------------------------------
    def set_profiling(self, enable: bool, max_launches: int = 8192) -> None:
        if self._enc is None:
            raise RuntimeError("run one forward first (the encoder handle is created lazily)")
        capi.check(capi.load().eec_encoder_set_profiling(self._enc, int(enable), max_launches), "set_profiling")

    def read_profile(self) -> Dict[str, Tuple[float, int]]:
        """{kernel class: (total ms, launches)} of the launches recorded since set_profiling(True)."""
        n = len(capi.KERNEL_CLASSES)
        ms, cnt = (C.c_double * n)(), (C.c_longlong * n)()
        capi.check(capi.load().eec_encoder_profile_read(self._enc, ms, cnt, n), "profile_read")
        return {k: (ms[i], cnt[i]) for i, k in enumerate(capi.KERNEL_CLASSES)}

    # -- data-parallel training (BASELINE.json configs[3]) -----------------------
    def enable_data_parallel(self, b_local: int, group=None, min_bucket_bytes: int = 4 << 20) -> None:
        """One process per GPU, every rank holding ``b_local`` utterances of the global batch: from now on the training
        backward writes the gradients into flat per-exit-group buckets (``parallel.GradBuckets``: ``p.grad`` become views
        of them) and, as ``eec_train_backward_ex`` reports each finished exit group, starts that bucket's all-reduce
        (weighted b_local / global batch: the loss is a batch mean, train.py:60-65) on the backend's stream, under the
        backward of the earlier groups.  Call ``sync_gradients()`` after ``loss.backward()`` and before clipping / the
        optimizer step.  Collective at set-up: one exchange of the shard sizes.  A no-op without an initialised process
        group of more than one rank (the buckets are still used, so the single-GPU step runs the same code)."""
        from . import parallel
        dev = next(self.parameters()).device
        named = [(n, p) for n, p in self.named_parameters()]
        self._dp = {"buckets": parallel.GradBuckets(named, self._cfg.n_exits, min_bucket_bytes=min_bucket_bytes),
                    "weight": parallel.shard_weight(b_local, dev, group), "group": group, "reduced": set(),
                    "active": parallel._active(group)}

    def sync_gradients(self) -> int:
        """Join the gradient collectives the last backward started from its progress callback and reduce EVERY other bucket
        now: buckets whose gradients the backward could not write into the flat views (gradients accumulated into existing
        ``.grad`` tensors, autograd-owned decoder gradients) and every bucket of a model whose backward does not report
        into the buckets at all (Splitformer / Early_zipformer / the heads-only step: their autograd functions know nothing
        of ``_dp``).  Call exactly once per backward, before clipping / the optimizer step.  Returns the number of
        collectives joined."""
        dp = getattr(self, "_dp", None)
        if dp is None or not dp["active"]:
            return 0
        buckets, early = dp["buckets"], dp["reduced"]
        for i in range(len(buckets.buckets)):
            if i not in early:
                buckets.allreduce_bucket(i, dp["weight"], dp["group"])
        n = buckets.wait()
        # a bucket reduced from the callback was reduced BEFORE autograd installed its views as p.grad: if autograd kept a
        # copy instead (a tensor hook, another live reference to the view), that copy was taken from an unreduced buffer
        for i in early:
            buckets.adopt_views(i)
        dp["reduced"] = set()
        return n

    # -- forward ------------------------------------------------------------
    def _workspace(self, B: int, T: int, device: torch.device) -> Tensor:
        k = (B, T, device.index or 0)
        ws = self._ws.get(k)
        if ws is None:
            n = capi.load().eec_encoder_workspace_bytes(self._enc, B, T)
            if len(self._ws) > 4:
                self._ws.clear()
            ws = torch.empty(n + 256, dtype=torch.uint8, device=device)
            self._ws[k] = ws
        return ws

    def _run_encoder(self, src: Tensor, lengths: Tensor, want_out: bool = True, want_taps: bool = False,
                     stop_after: int = -1, want_x: bool = False, n_groups: Optional[int] = None):
        """``n_groups`` (1 .. E): stop after that many exit groups (eec_encoder_forward_prefix, production launch plan);
        ``out`` / ``taps`` then hold only the exits that were run."""
        if not src.is_cuda:
            raise RuntimeError("the MI355X encoder runs on a HIP device only; move the model and inputs to "
                               "'cuda' (there is no CPU fallback -- the CPU reference lives in oracle/).")
        if self.training and torch.is_grad_enabled() and not getattr(self, "_frozen_encoder_pass", False):
            raise NotImplementedError("this class has no training step on the HIP path (Early_conformer and full_conformer do); "
                                      "call under model.eval() / torch.no_grad()")
        if src.dim() != 3 or src.size(1) != self._cfg.n_mels:
            raise ValueError(f"src must be [B, {self._cfg.n_mels}, T], got {tuple(src.shape)}")
        dev = src.device
        with torch.cuda.device(dev):
            self._ensure_packed(dev)
            lib = capi.load()
            B, _, T = src.shape
            Tq = lib.eec_out_frames(T)
            if Tq <= 0:
                raise ValueError("T too short for two k=3 s=2 convolutions")
            src = src.contiguous().float()
            len_dev = _to_device(lengths, dev)
            E, D, V = self._cfg.n_exits, self._cfg.d_model, self._cfg.vocab
            if n_groups is not None:
                if stop_after >= 0:
                    raise ValueError("n_groups and stop_after are exclusive")
                if not 1 <= int(n_groups) <= E:
                    raise ValueError(f"n_groups must be in 1 .. {E}")
                E = int(n_groups)
            out = torch.empty((E, B, Tq, V), dtype=torch.float32, device=dev) if want_out else None
            taps = torch.empty((E, B, Tq, D), dtype=torch.float32, device=dev) if want_taps else None
            xdbg = torch.empty((B, Tq, D), dtype=torch.float32, device=dev) if want_x else None
            ws = self._workspace(B, T, dev)
            ws_ptr = (ws.data_ptr() + 255) // 256 * 256
            stream = torch.cuda.current_stream(dev).cuda_stream
            ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
            if n_groups is not None:
                rc = lib.eec_encoder_forward_prefix(
                    self._enc, src.data_ptr(), len_dev.data_ptr(), B, T, capi.PRECISIONS[self.precision], E,
                    ptr(out), ptr(taps), ptr(xdbg), ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()), C.c_void_p(stream))
                capi.check(rc, "eec_encoder_forward_prefix")
            else:
                rc = lib.eec_encoder_forward(
                    self._enc, src.data_ptr(), len_dev.data_ptr(), B, T, capi.PRECISIONS[self.precision],
                    ptr(out), ptr(taps), ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()), stop_after, ptr(xdbg),
                    C.c_void_p(stream))
                capi.check(rc, "eec_encoder_forward")
            # src/len_dev must outlive the asynchronous launches on this stream
            src.record_stream(torch.cuda.current_stream(dev))
            len_dev.record_stream(torch.cuda.current_stream(dev))
        return out, taps, xdbg


class Early_conformer(_HipEncoderMixin, nn.Module):
    """CTC early-exit Conformer; ``forward(src[B,n_mels,T], lengths[B]) -> [E,B,T',V]`` log-probs."""

    def __init__(self, src_pad_idx, n_enc_exits, enc_voc_size, dec_voc_size, d_model, n_head, max_len,
                 d_feed_forward, n_enc_layers, features_length, drop_prob, depthwise_kernel_size, device=None):
        nn.Module.__init__(self)
        self.input_dim, self.num_heads, self.ffn_dim = d_model, n_head, d_feed_forward
        self.num_layers, self.depthwise_conv_kernel_size = n_enc_layers, depthwise_kernel_size
        self.n_enc_exits, self.dropout, self.device, self.src_pad_idx = n_enc_exits, drop_prob, device, src_pad_idx
        self.conv_subsample = Conv1dSubampling(features_length, d_model)
        self.positional_encoder = PositionalEncoding(d_model, drop_prob, max_len)
        self.linears = nn.ModuleList([nn.Linear(d_model, dec_voc_size) for _ in range(n_enc_exits)])
        self.conformer = nn.ModuleList([
            Conformer(input_dim=d_model, num_heads=n_head, ffn_dim=d_feed_forward, num_layers=n_enc_layers,
                      depthwise_conv_kernel_size=depthwise_kernel_size, dropout=drop_prob)
            for _ in range(n_enc_exits)])
        self._hip_init(d_model, n_head, d_feed_forward, depthwise_kernel_size, n_enc_exits, n_enc_layers,
                       features_length, dec_voc_size, max_len)

    def forward(self, src: Tensor, lengths: Tensor) -> Tensor:
        if self.training and type(self) is Early_conformer:
            # train-mode semantics (batch-statistics BatchNorm, dropout, running-statistics update) whenever the module is in
            # train mode -- with or without autograd, whatever requires_grad says, as the reference.  The one exception is an
            # explicit opt-in: ``model.frozen_encoder_eval = True`` runs a FROZEN encoder (only linears.* trainable) on the fused
            # inference path in eval semantics and trains the heads on its taps (cheaper; not what the reference computes).
            frozen = not any(p.requires_grad for n, p in self.named_parameters() if not n.startswith("linears."))
            if torch.is_grad_enabled() and frozen and getattr(self, "frozen_encoder_eval", False):
                return self._forward_heads_trainable(src, lengths)
            return self._forward_train(src, lengths)
        if self.training and torch.is_grad_enabled():
            return self._forward_heads_trainable(src, lengths)
        return self._run_encoder(src, lengths)[0]

    def _forward_train(self, src: Tensor, lengths: Tensor, want_taps: bool = False):
        """The training step's forward (train.py:54) on the HIP training kernels; autograd reaches every parameter of the
        path (stem, Conformer groups, exit heads).  ``want_taps``: also return the group outputs [E, B, T', D] as a second
        differentiable result (what full_conformer hands to its attention decoders)."""
        if not src.is_cuda:
            raise RuntimeError("the MI355X training step runs on a HIP device only (there is no CPU fallback)")
        if src.dim() != 3 or src.size(1) != self._cfg.n_mels:
            raise ValueError(f"src must be [B, {self._cfg.n_mels}, T], got {tuple(src.shape)}")
        mine = ("conv_subsample.", "conformer.", self._head_attr + ".")
        named = [(n, p) for n, p in _named_tensors(self)[0] if n.startswith(mine)]
        names = tuple(n for n, _ in named)
        len_dev = _to_device(lengths, src.device)
        return _EncoderTrainFn.apply(self, src.contiguous().float(), len_dev, names, want_taps, *[p for _, p in named])

    def _forward_heads_trainable(self, src: Tensor, lengths: Tensor) -> Tensor:
        """First slice of the training path (train.py:53-70): the exit heads ``linears.*`` are trainable on a FROZEN
        encoder.  The encoder stack runs on the HIP path without autograd, in eval semantics (running BatchNorm
        statistics, no dropout) whatever ``self.training`` says; the heads are an autograd function over its taps, so
        ``exit_ctc_losses(model(src, lengths), ...).sum().backward()`` fills ``linears.*.grad``.  With a trainable parameter
        anywhere else ``Early_conformer.forward`` takes the full training step instead (``_forward_train``); the classes that
        reuse this method (Splitformer, Early_zipformer, full_conformer) still raise for those."""
        trainable = [n for n, p in self.named_parameters() if p.requires_grad and not n.startswith("linears.")]
        if trainable:
            raise NotImplementedError("this class trains its exit heads (linears.*) only: freeze the encoder "
                                      f"(requires_grad_(False)); trainable now: {trainable[:3]}{' ...' if len(trainable) > 3 else ''}")
        self._frozen_encoder_pass = True
        try:
            with torch.no_grad():
                taps = self._run_encoder(src, lengths, want_out=False, want_taps=True, n_groups=self._cfg.n_exits)[1]
        finally:
            self._frozen_encoder_pass = False
        wb = [l.weight for l in self.linears] + [l.bias for l in self.linears]
        return _ExitHeadsFn.apply(self, taps, *wb)

    def forward_exits(self, src: Tensor, lengths: Tensor, n_exits: int) -> Tensor:
        """Early exit (extension; the reference's forward always runs every group): log-probs of the first
        ``n_exits`` exits only, [n_exits, B, T', V], at n_exits / E of the cost of ``forward``."""
        return self._run_encoder(src, lengths, n_groups=n_exits)[0]

    def greedy_decode(self, enc_out: Tensor, blank: int = 0) -> List[List[List[int]]]:
        """Batched GreedyCTCDecoder (util/beam_infer.py:9-24) over every exit and utterance."""
        E, B, Tq, V = enc_out.shape
        tokens, counts = greedy_ctc(enc_out.reshape(E * B, Tq, V), blank)
        tokens, counts = tokens.cpu(), counts.cpu()
        return [[tokens[e * B + b, : counts[e * B + b]].tolist() for b in range(B)] for e in range(E)]


# ---- building blocks of the training step (Splitformer / Early_zipformer: train.py:180-208) ----------------------------------
_GROUP_FIELDS = [f for f in capi._LAYER_FIELDS if f not in ("conv_bn_rm", "conv_bn_rv")]  # the 30 trainable tensors of a ConformerLayer


def _group_layer_tensors(group: nn.Module) -> List[Tensor]:
    """The parameters of a Conformer group, layer-major, in _GROUP_FIELDS order."""
    out: List[Tensor] = []
    for layer in group.conformer_layers:
        sd = dict(layer.named_parameters())
        out += [sd[capi.LAYER_KEYS[f]] for f in _GROUP_FIELDS]
    return out


def _group_struct(tensors: Sequence[Tensor], n_layers: int):
    layers = (capi.EecLayerParams * n_layers)()
    k = len(_GROUP_FIELDS)
    for l in range(n_layers):
        for i, f in enumerate(_GROUP_FIELDS):
            setattr(layers[l], f, tensors[l * k + i].data_ptr())
    return layers


def _aligned_ws(nbytes: int, dev) -> Tuple[Tensor, int]:
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    return ws, (ws.data_ptr() + 255) // 256 * 256


class _TrainGroupFn(torch.autograd.Function):
    """One Conformer group (torchaudio ``Conformer(num_layers=L)``: early_exit.py:160-172, 266-297) in train mode on rows
    x [B, T', D] with key lengths key_len [B] (int32, device), and its backward, on the HIP training kernels
    (eec_train_group_forward / _backward).  BatchNorm uses the batch statistics and updates the running ones like nn.BatchNorm1d."""

    @staticmethod
    def forward(ctx, model, group, x, key_len, seed, site_base, *params):
        lib = capi.load()
        dev = x.device
        cfg = model._cfg
        B, Tq, D = x.shape
        L = len(group.conformer_layers)
        for t in params:
            if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError(f"group parameters must be contiguous fp32 tensors on {dev}")
        x = x.contiguous().float()
        with torch.cuda.device(dev):
            layers = _group_struct(params, L)
            nbytes = lib.eec_train_group_workspace_bytes(C.byref(cfg), L, B, Tq)
            if nbytes == 0:
                raise ValueError("unsupported geometry for a training group")
            ws, ws_ptr = _aligned_ws(nbytes, dev)
            out = torch.empty_like(x)
            bn = torch.empty((L, 2, D), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _trainer_check(lib.eec_train_group_forward(C.byref(cfg), layers, L, x.data_ptr(), key_len.data_ptr(), B, Tq, int(model.train_passes),
                                                       float(model.dropout), int(seed), int(site_base), out.data_ptr(), bn.data_ptr(), ws_ptr, nbytes,
                                                       C.c_void_p(stream)), "eec_train_group_forward")
            n = B * Tq
            with torch.no_grad():
                for l, layer in enumerate(group.conformer_layers):
                    bnm = layer.conv_module.sequential[3]
                    if bnm.track_running_stats and bnm.running_mean is not None:
                        m = bnm.momentum if bnm.momentum is not None else 0.1
                        bnm.running_mean.mul_(1 - m).add_(bn[l, 0], alpha=m)
                        bnm.running_var.mul_(1 - m).add_(bn[l, 1] * (n / max(n - 1, 1)), alpha=m)
                        bnm.num_batches_tracked += 1
        ctx.model, ctx.L, ctx.seed, ctx.site_base = model, L, int(seed), int(site_base)
        ctx.ws, ctx.ws_ptr, ctx.nbytes = ws, ws_ptr, nbytes
        ctx.passes, ctx.drop = int(model.train_passes), float(model.dropout)
        ctx.save_for_backward(x, key_len, *params)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if ctx.ws is None:
            raise RuntimeError("this group's recorded forward was already consumed by a backward")
        x, key_len, params = ctx.saved_tensors[0], ctx.saved_tensors[1], ctx.saved_tensors[2:]
        dev = x.device
        lib = capi.load()
        B, Tq, _ = x.shape
        g = g.contiguous().float()
        with torch.cuda.device(dev):
            layers = _group_struct(params, ctx.L)
            grads = [torch.empty_like(t) for t in params]
            glayers = _group_struct(grads, ctx.L)
            g_in = torch.empty_like(x)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _trainer_check(lib.eec_train_group_backward(C.byref(ctx.model._cfg), layers, glayers, ctx.L, x.data_ptr(), key_len.data_ptr(), B, Tq,
                                                        ctx.passes, ctx.drop, ctx.seed, ctx.site_base, g.data_ptr(), g_in.data_ptr(), ctx.ws_ptr,
                                                        ctx.nbytes, C.c_void_p(stream)), "eec_train_group_backward")
        ctx.ws = None
        need = ctx.needs_input_grad[6:]
        return (None, None, g_in if ctx.needs_input_grad[2] else None, None, None, None, *[gr if nd else None for gr, nd in zip(grads, need)])


class _TrainStemFn(torch.autograd.Function):
    """Stem in train mode: Conv1d(k3, s2) [-> Conv1d(k3, s2)] -> + positional encoding -> dropout (early_exit.py:24-48 / 80-95,
    positional_encoding.py:65-73) -> [B, To, D]; no gradient with respect to the mel input."""

    @staticmethod
    def forward(ctx, model, mel, pe, seed, site, w0, b0, w1, b1):
        lib = capi.load()
        dev = mel.device
        cfg = model._cfg
        B, _, T = mel.shape
        two = w1 is not None
        T1 = (T - 3) // 2 + 1
        To = ((T1 - 3) // 2 + 1) if two else T1
        with torch.cuda.device(dev):
            nbytes = lib.eec_train_stem_workspace_bytes(C.byref(cfg), B, T, int(two))
            if nbytes == 0:
                raise ValueError("unsupported geometry for the training stem")
            ws, ws_ptr = _aligned_ws(nbytes, dev)
            out = torch.empty((B, To, cfg.d_model), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _trainer_check(lib.eec_train_stem_forward(C.byref(cfg), w0.data_ptr(), b0.data_ptr(), w1.data_ptr() if two else None,
                                                      b1.data_ptr() if two else None, pe.data_ptr(), mel.data_ptr(), B, T, int(model.train_passes),
                                                      float(model.dropout), int(seed), int(site), out.data_ptr(), ws_ptr, nbytes, C.c_void_p(stream)),
                           "eec_train_stem_forward")
        ctx.model, ctx.geo, ctx.two = model, (B, T), two
        ctx.seed, ctx.site, ctx.passes, ctx.drop = int(seed), int(site), int(model.train_passes), float(model.dropout)
        ctx.ws, ctx.ws_ptr, ctx.nbytes = ws, ws_ptr, nbytes
        ctx.save_for_backward(mel, w0, b0, *((w1, b1) if two else ()))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        saved = ctx.saved_tensors
        w0, b0 = saved[1], saved[2]
        dev = g.device
        lib = capi.load()
        B, T = ctx.geo
        g = g.contiguous().float()
        with torch.cuda.device(dev):
            g_w0, g_b0 = torch.empty_like(w0), torch.empty_like(b0)
            g_w1 = torch.empty_like(saved[3]) if ctx.two else None
            g_b1 = torch.empty_like(saved[4]) if ctx.two else None
            stream = torch.cuda.current_stream(dev).cuda_stream
            _trainer_check(lib.eec_train_stem_backward(C.byref(ctx.model._cfg), int(ctx.two), B, T, ctx.passes, ctx.drop, ctx.seed, ctx.site,
                                                       g.data_ptr(), g_w0.data_ptr(), g_b0.data_ptr(), g_w1.data_ptr() if ctx.two else None,
                                                       g_b1.data_ptr() if ctx.two else None, ctx.ws_ptr, ctx.nbytes, C.c_void_p(stream)),
                           "eec_train_stem_backward")
        ctx.ws = None
        return (None, None, None, None, None, g_w0, g_b0, g_w1, g_b1)


class _TrainHeadFn(torch.autograd.Function):
    """Exit head ``log_softmax(x . W^T + b)`` (early_exit.py:629-631) and its backward on the training GEMM."""

    @staticmethod
    def forward(ctx, passes, x, W, b):
        lib = capi.load()
        dev = x.device
        x = x.contiguous().float()
        M, D = x.shape
        V = W.size(0)
        with torch.cuda.device(dev):
            logp = torch.empty((M, V), dtype=torch.float32, device=dev)
            scratch = torch.empty((M, V), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _trainer_check(lib.eec_train_head_forward(x.data_ptr(), W.data_ptr(), b.data_ptr(), M, V, D, int(passes), logp.data_ptr(),
                                                      scratch.data_ptr(), C.c_void_p(stream)), "eec_train_head_forward")
            scratch.record_stream(torch.cuda.current_stream(dev))
        ctx.passes = int(passes)
        ctx.save_for_backward(x, W, logp)
        return logp

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, W, logp = ctx.saved_tensors
        lib = capi.load()
        dev = x.device
        M, D = x.shape
        V = W.size(0)
        g = g.contiguous().float()
        with torch.cuda.device(dev):
            dW, db = torch.empty_like(W), torch.empty((V,), dtype=torch.float32, device=dev)
            dx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
            scratch = torch.empty((lib.eec_train_head_backward_scratch_floats(M, V, D),), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _trainer_check(lib.eec_train_head_backward(x.data_ptr(), W.data_ptr(), logp.data_ptr(), g.data_ptr(), M, V, D, ctx.passes,
                                                       dx.data_ptr() if dx is not None else None, dW.data_ptr(), db.data_ptr(), scratch.data_ptr(),
                                                       C.c_void_p(stream)), "eec_train_head_backward")
            scratch.record_stream(torch.cuda.current_stream(dev))
        return (None, dx, dW, db)


def _train_group(model, group: nn.Module, x: Tensor, key_len: Tensor, seed: int, site_base: int) -> Tensor:
    return _TrainGroupFn.apply(model, group, x, key_len, seed, site_base, *_group_layer_tensors(group))


def _train_head(model, linear: nn.Linear, x: Tensor) -> Tensor:
    B, Tq, D = x.shape
    return _TrainHeadFn.apply(model.train_passes, x.reshape(B * Tq, D), linear.weight, linear.bias).reshape(B, Tq, -1)


class _TimeResample(nn.Module):
    """Parameterless stand-ins that keep the reference's module tree (Downsampling / Upsampling, early_exit.py:95-114)."""

    def __init__(self, factor: int):
        super().__init__()
        self.factor = factor


class Splitformer(Early_conformer):
    """Drop-in for the reference's ``Splitformer`` (early_exit.py:227-364): Early_conformer plus,
    at the first and the last exit, a one-layer Conformer on the 2x time-down-sampled input of that exit group, added
    back (nearest-neighbour up-sampled) before the head.  Same constructor kwargs, ``forward(src, lengths)`` and
    state_dict names.  Every Conformer group and every head runs in libeec (eec_encoder_group_forward /
    eec_encoder_head_forward, the production chain-kernel plan); the strided slice, the repeat and the add that glue the
    branch in are three torch ops on [B, T', 256] tensors.  Reference quirks kept: the branch's key lengths are
    ``clamp((mel_lengths + pad) / 2, max=T'/2)`` (:324-331) and ``index // (n_enc_exits - 1)`` picks the branch."""

    factor = 2

    def __init__(self, src_pad_idx, n_enc_exits, enc_voc_size, dec_voc_size, d_model, n_head, max_len,
                 d_feed_forward, n_enc_layers, features_length, drop_prob, depthwise_kernel_size, device=None):
        if n_enc_exits < 2:
            raise ValueError("Splitformer needs n_enc_exits >= 2 (the reference divides by n_enc_exits - 1)")
        super().__init__(src_pad_idx, n_enc_exits, enc_voc_size, dec_voc_size, d_model, n_head, max_len,
                         d_feed_forward, n_enc_layers, features_length, drop_prob, depthwise_kernel_size, device)
        self.downsampling = nn.ModuleList([_TimeResample(self.factor) for _ in range(2)])
        self.upsampling = nn.ModuleList([_TimeResample(self.factor) for _ in range(2)])
        self.conformer_parallel = nn.ModuleList([
            Conformer(input_dim=d_model, num_heads=n_head, ffn_dim=d_feed_forward, num_layers=1,
                      depthwise_conv_kernel_size=depthwise_kernel_size, dropout=drop_prob) for _ in range(2)])
        # second libeec handle: the two one-layer branch groups, packed without stem and heads
        self._par_cfg = capi.EecConfig(d_model, n_head, d_feed_forward, depthwise_kernel_size, 2, 1, features_length,
                                       dec_voc_size, max_len, capi.ARCH_CONFORMER)
        self._par_enc = None
        self._par_device = None
        self._par_key = None

    def __del__(self):
        enc = getattr(self, "_par_enc", None)
        if enc is not None:
            try:
                capi.load().eec_encoder_destroy(enc)
            except Exception:
                pass
        super().__del__()

    def _ensure_branch_packed(self, device: torch.device) -> None:
        tensors = list(self.conformer_parallel.parameters()) + list(self.conformer_parallel.buffers())
        key = (device, tuple(t._version for t in tensors), tuple(t.data_ptr() for t in tensors))
        if self._par_enc is not None and key == self._par_key:
            return
        lib = capi.load()
        if self._par_enc is not None and self._par_device != device:
            lib.eec_encoder_destroy(self._par_enc)
            self._par_enc, self._par_key = None, None
        if self._par_enc is None:
            h = C.c_void_p()
            capi.check(lib.eec_encoder_create(C.byref(self._par_cfg), C.byref(h)), "eec_encoder_create")
            self._par_enc, self._par_device = h, device
        sd = dict(self.state_dict(keep_vars=True))

        def ptr(name: str) -> int:
            t = sd[name]
            if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError(f"parameter {name} must be a contiguous fp32 tensor on {device}")
            return t.data_ptr()

        params = capi.EecParams(None, None, None, None, None, _layer_params(ptr, "conformer_parallel", 2, 1), None, None)
        stream = torch.cuda.current_stream(device).cuda_stream
        capi.check(lib.eec_encoder_pack(self._par_enc, C.byref(params), C.c_void_p(stream)), "eec_encoder_pack")
        self._par_key = key

    def _forward_training(self, src: Tensor, lengths: Tensor) -> Tensor:
        """train.py:180-208 (--model_type splitformer) in train mode: every module on the HIP training kernels behind autograd
        functions (stem, Conformer groups -- the E main ones and the two down-sampled branches --, heads); the strided slice, the
        repeat and the add that glue the branches in are the same torch ops as in inference, differentiated by autograd."""
        if not src.is_cuda:
            raise RuntimeError("the MI355X training step runs on a HIP device only (there is no CPU fallback)")
        dev = src.device
        E = self._cfg.n_exits
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        conv = self.conv_subsample.sequential
        x = _TrainStemFn.apply(self, src.contiguous().float(), self.positional_encoder.pe, seed, 1, conv[0].weight, conv[0].bias,
                               conv[1].weight, conv[1].bias)
        B, Tq, D = x.shape
        mel_len = _to_device(lengths, dev)
        base = torch.clamp(mel_len / 4, max=Tq).to(torch.int32)
        outs = []
        for index in range(E):
            branch = index in (0, E - 1)
            side = x
            x = _train_group(self, self.conformer[index], x, base, seed, 16 + 128 * index)
            if branch:
                pad = (-Tq) % self.factor
                if pad:
                    side = torch.cat((side, side.new_zeros(B, pad, D)), dim=1)
                side = side[:, :: self.factor, :].contiguous()
                side_len = torch.clamp((mel_len + pad) / self.factor, max=side.size(1)).to(torch.int32)
                side = _train_group(self, self.conformer_parallel[index // (E - 1)], side, side_len, seed, 16 + 128 * index + 64)
                x = x + torch.repeat_interleave(side, self.factor, dim=1)[:, :Tq, :]
            outs.append(_train_head(self, self.linears[index], x))
        return torch.stack(outs)

    def forward(self, src: Tensor, lengths: Tensor) -> Tensor:
        if self.training:
            return self._forward_training(src, lengths)
        # stem (+ PE) through the monolithic entry's first sub-step; also validates src and packs the main handle
        x = self._run_encoder(src, lengths, want_out=False, stop_after=0, want_x=True)[2]
        dev = x.device
        with torch.cuda.device(dev):
            self._ensure_branch_packed(dev)
            lib = capi.load()
            B, Tq, D = x.shape
            E, V = self._cfg.n_exits, self._cfg.vocab
            mel_len = _to_device(lengths, dev)
            base = torch.clamp(mel_len / 4, max=Tq).to(torch.int32)
            out = torch.empty((E, B, Tq, V), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            for index in range(E):
                branch = index in (0, E - 1)
                side = x.clone() if branch else None  # the group's INPUT feeds the branch; the group runs in place
                self._group(self._enc, index, x, base)
                if branch:
                    pad = (-Tq) % self.factor
                    if pad:
                        side = torch.cat((side, side.new_zeros(B, pad, D)), dim=1)
                    side = side[:, :: self.factor, :].contiguous()
                    side_len = torch.clamp((mel_len + pad) / self.factor, max=side.size(1)).to(torch.int32)
                    self._group(self._par_enc, index // (E - 1), side, side_len)
                    x = x + torch.repeat_interleave(side, self.factor, dim=1)[:, :Tq, :]
                rc = lib.eec_encoder_head_forward(self._enc, index, x.data_ptr(), B * Tq, out[index].data_ptr(),
                                                  capi.PRECISIONS[self.precision], C.c_void_p(stream))
                capi.check(rc, "eec_encoder_head_forward")
        return out


class Conv1dSubampling_Zipformer(nn.Module):
    """Parameter holder for the one-convolution stem (early_exit.py:80-95)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=3, stride=2, padding=0)


class Early_zipformer(_HipEncoderMixin, nn.Module):
    """Drop-in for the reference's ``Early_zipformer`` (early_exit.py:117-224): one-convolution stem,
    two Conformer groups at full frame rate, five stacks of groups at 1/2, 1/4, 1/8, 1/4, 1/2 rate with a skip around
    each stack, one head on every second frame -> [1, B, ceil(T1/2), V].  Same kwargs, ``forward(src, lengths)`` and
    state_dict names; ``n_enc_exits`` is the number of Conformer groups and must be >= 19 (the forward indexes groups
    0 .. 18).  Stem, every group and the head run in libeec; pad / stride / repeat / add are torch ops."""

    factors = (2, 4, 8, 4, 2)
    stack = (2, 4, 5, 4, 2)

    def __init__(self, src_pad_idx, n_enc_exits, enc_voc_size, dec_voc_size, d_model, n_head, max_len,
                 d_feed_forward, n_enc_layers, features_length, drop_prob, depthwise_kernel_size, device=None):
        nn.Module.__init__(self)
        if n_enc_exits < 2 + sum(self.stack):
            raise ValueError(f"Early_zipformer indexes Conformer groups 0 .. {1 + sum(self.stack)}: n_enc_exits must be >= "
                             f"{2 + sum(self.stack)}")
        self.n_enc_exits, self.dropout, self.device, self.src_pad_idx = n_enc_exits, drop_prob, device, src_pad_idx
        self.downsampling = nn.ModuleList([_TimeResample(f) for f in self.factors])
        self.downsampling_output = _TimeResample(2)
        self.upsampling = nn.ModuleList([_TimeResample(f) for f in self.factors])
        self.conv_subsample = Conv1dSubampling_Zipformer(features_length, d_model)
        self.positional_encoder = PositionalEncoding(d_model, drop_prob, max_len)
        self.linear = nn.Linear(d_model, dec_voc_size)
        self.conformer = nn.ModuleList([
            Conformer(input_dim=d_model, num_heads=n_head, ffn_dim=d_feed_forward, num_layers=n_enc_layers,
                      depthwise_conv_kernel_size=depthwise_kernel_size, dropout=drop_prob)
            for _ in range(n_enc_exits)])
        self._hip_init(d_model, n_head, d_feed_forward, depthwise_kernel_size, n_enc_exits, n_enc_layers,
                       features_length, dec_voc_size, max_len)

    def _param_tensors(self) -> List[Tensor]:
        return list(self.conv_subsample.parameters()) + list(self.conformer.parameters()) + \
            list(self.conformer.buffers()) + list(self.linear.parameters()) + [self.positional_encoder.pe]

    def _pack(self, lib, device, sd, ptr) -> None:
        E, L = self._cfg.n_exits, self._cfg.layers_per_exit
        hw = (C.c_void_p * E)(*[ptr("linear.weight")] * E)  # one head, registered under every exit index
        hb = (C.c_void_p * E)(*[ptr("linear.bias")] * E)
        params = capi.EecParams(ptr("conv_subsample.conv.weight"), ptr("conv_subsample.conv.bias"), None, None,
                                ptr("positional_encoder.pe"), _layer_params(ptr, "conformer", E, L), hw, hb)
        stream = torch.cuda.current_stream(device).cuda_stream
        capi.check(lib.eec_encoder_pack(self._enc, C.byref(params), C.c_void_p(stream)), "eec_encoder_pack")

    def forward(self, src: Tensor, lengths: Tensor) -> Tensor:
        if not src.is_cuda:
            raise RuntimeError("the MI355X encoder runs on a HIP device only (there is no CPU fallback -- the CPU "
                               "reference lives in oracle/).")
        if src.dim() != 3 or src.size(1) != self._cfg.n_mels or src.size(2) < 3:
            raise ValueError(f"src must be [B, {self._cfg.n_mels}, T >= 3], got {tuple(src.shape)}")
        if self.training:
            return self._forward_training(src, lengths)
        dev = src.device
        with torch.cuda.device(dev):
            self._ensure_packed(dev)
            lib = capi.load()
            src = src.contiguous().float()
            B, _, T = src.shape
            T1, D, V = (T - 3) // 2 + 1, self._cfg.d_model, self._cfg.vocab
            stream = torch.cuda.current_stream(dev).cuda_stream
            enc = torch.empty((B, T1, D), dtype=torch.float32, device=dev)
            capi.check(lib.eec_encoder_stem1_forward(self._enc, src.data_ptr(), B, T, enc.data_ptr(), C.c_void_p(stream)),
                       "eec_encoder_stem1_forward")
            mel_len = _to_device(lengths, dev)
            base = torch.clamp(mel_len / 2, max=T1).to(torch.int32)
            self._group(self._enc, 0, enc, base)
            self._group(self._enc, 1, enc, base)
            first = 2
            for factor, count in zip(self.factors, self.stack):
                skip = enc
                n = enc.size(1)
                pad = (-n) % factor
                if pad:
                    enc = torch.cat((enc, enc.new_zeros(B, pad, D)), dim=1)
                enc = enc[:, ::factor, :].contiguous()  # a fresh tensor: the groups below run in place
                key_len = torch.clamp((mel_len + pad) / factor, max=enc.size(1)).to(torch.int32)
                for g in range(first, first + count):
                    self._group(self._enc, g, enc, key_len)
                first += count
                enc = torch.repeat_interleave(enc, factor, dim=1)[:, :n, :] + skip
            rows = enc[:, ::2, :].contiguous()
            out = torch.empty((1, B, rows.size(1), V), dtype=torch.float32, device=dev)
            rc = lib.eec_encoder_head_forward(self._enc, 0, rows.data_ptr(), B * rows.size(1), out.data_ptr(),
                                              capi.PRECISIONS[self.precision], C.c_void_p(stream))
            capi.check(rc, "eec_encoder_head_forward")
            src.record_stream(torch.cuda.current_stream(dev))
        return out


def _zipformer_forward_training(self, src: Tensor, lengths: Tensor) -> Tensor:
    """train.py:180-208 (--model_type zipformer) in train mode: one-convolution stem, the 19 Conformer groups at five frame rates
    and the head on the HIP training kernels behind autograd functions; pad / stride / repeat / add are torch ops under autograd."""
    dev = src.device
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    conv = self.conv_subsample.conv
    enc = _TrainStemFn.apply(self, src.contiguous().float(), self.positional_encoder.pe, seed, 1, conv.weight, conv.bias, None, None)
    B, T1, D = enc.shape
    mel_len = _to_device(lengths, dev)
    base = torch.clamp(mel_len / 2, max=T1).to(torch.int32)
    enc = _train_group(self, self.conformer[0], enc, base, seed, 16)
    enc = _train_group(self, self.conformer[1], enc, base, seed, 16 + 128)
    first = 2
    for factor, count in zip(self.factors, self.stack):
        skip = enc
        n = enc.size(1)
        pad = (-n) % factor
        if pad:
            enc = torch.cat((enc, enc.new_zeros(B, pad, D)), dim=1)
        enc = enc[:, ::factor, :].contiguous()
        key_len = torch.clamp((mel_len + pad) / factor, max=enc.size(1)).to(torch.int32)
        for g in range(first, first + count):
            enc = _train_group(self, self.conformer[g], enc, key_len, seed, 16 + 128 * g)
        first += count
        enc = torch.repeat_interleave(enc, factor, dim=1)[:, :n, :] + skip
    rows = enc[:, ::2, :].contiguous()
    return _train_head(self, self.linear, rows).unsqueeze(0)


Early_zipformer._forward_training = _zipformer_forward_training


def encoder_lengths(lengths: Tensor, t_out: int) -> Tensor:
    """``clamp(lengths / 4, max=T').to(int)`` (early_exit.py:623) on the device: int64 [B] -> int32 [B]."""
    if not lengths.is_cuda:
        raise RuntimeError("encoder_lengths runs on a HIP device only")
    lengths = lengths.to(torch.int64).contiguous()
    out = torch.empty((lengths.numel(),), dtype=torch.int32, device=lengths.device)
    with torch.cuda.device(lengths.device):
        stream = torch.cuda.current_stream(lengths.device).cuda_stream
        capi.check(capi.load().eec_encoder_lengths(lengths.data_ptr(), lengths.numel(), int(t_out), out.data_ptr(),
                                                   C.c_void_p(stream)), "eec_encoder_lengths")
    return out


def greedy_ctc(logp: Tensor, blank: int = 0) -> Tuple[Tensor, Tensor]:
    """[N, T', V] fp32 log-probs on the GPU -> (tokens [N, T'] int32, counts [N] int32)."""
    if not logp.is_cuda:
        raise RuntimeError("greedy_ctc runs on a HIP device only")
    logp = logp.contiguous().float()
    N, Tq, V = logp.shape
    tokens = torch.empty((N, Tq), dtype=torch.int32, device=logp.device)
    counts = torch.empty((N,), dtype=torch.int32, device=logp.device)
    with torch.cuda.device(logp.device):
        stream = torch.cuda.current_stream(logp.device).cuda_stream
        capi.check(capi.load().eec_greedy_ctc(logp.data_ptr(), N, Tq, V, blank, tokens.data_ptr(),
                                              counts.data_ptr(), C.c_void_p(stream)), "eec_greedy_ctc")
    return tokens, counts


def ctc_beam_decode(logp: Tensor, beam_size: int = 10, blank: int = 0, blank_skip_threshold: float = 0.95,
                    skip_drops_frame: bool = False):
    """CTC prefix beam search of [N, T', V] log-probs on the device (eec_ctc_beam_decode): the best hypothesis per
    sequence, as ``BeamInference.ctc_cuda_predict`` uses torchaudio's cuda_ctc_decoder (util/beam_infer.py:102-112).
    ``skip_drops_frame``: a frame above ``blank_skip_threshold`` is dropped instead of being taken as a blank frame (the two
    readings of the third-party decoder's skip rule, include/eec.h).  Returns (tokens [N, T'] int32, counts [N] int32,
    scores [N] fp32)."""
    if not logp.is_cuda:
        raise RuntimeError("ctc_beam_decode runs on a HIP device only")
    logp = logp.contiguous().float()
    N, Tq, V = logp.shape
    dev = logp.device
    lib = capi.load()
    tokens = torch.empty((N, Tq), dtype=torch.int32, device=dev)
    counts = torch.empty((N,), dtype=torch.int32, device=dev)
    scores = torch.empty((N,), dtype=torch.float32, device=dev)
    ws = torch.empty((lib.eec_ctc_beam_workspace_bytes(N, Tq),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        capi.check(lib.eec_ctc_beam_decode_ex(logp.data_ptr(), N, Tq, V, blank, beam_size, blank_skip_threshold, int(bool(skip_drops_frame)),
                                              ws.data_ptr(), tokens.data_ptr(), counts.data_ptr(), scores.data_ptr(), C.c_void_p(stream)),
                   "eec_ctc_beam_decode")
    return tokens, counts, scores


def _ctc_prepare(enc_out: Tensor, targets: Tensor, target_len: Tensor):
    if not enc_out.is_cuda:
        raise RuntimeError("exit_ctc_losses runs on a HIP device only")
    enc_out = enc_out.contiguous().float()
    dev = enc_out.device
    tg = _to_device(targets, dev)
    tl = _to_device(target_len, dev)
    return enc_out, tg, tl


class _ExitCtcLossFn(torch.autograd.Function):
    """Per-exit CTC losses [E] with their gradient with respect to the log-probs (eec_ctc_loss_forward / _backward):
    what autograd computes through the reference's loop of E nn.CTCLoss calls (train.py:60-68)."""

    @staticmethod
    def forward(ctx, enc_out, tg, tl, blank):
        E, B, Tq, V = enc_out.shape
        if V > 256 or V % 4:
            raise ValueError(f"exit_ctc_losses with a gradient needs a vocabulary of at most 256 entries, a multiple of 4 (got {V}): "
                             "the CTC gradient kernel holds a vocabulary row in one wave")
        dev = enc_out.device
        lib = capi.load()
        nll = torch.empty((E * B,), dtype=torch.float32, device=dev)
        out = torch.empty((E,), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.eec_ctc_backward_workspace_bytes(E, B, Tq, tg.size(1)) + 256,), dtype=torch.uint8, device=dev)
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            capi.check(lib.eec_ctc_loss_forward(enc_out.data_ptr(), tg.data_ptr(), tl.data_ptr(), E, B, Tq, V, tg.size(1), blank,
                                                nll.data_ptr(), out.data_ptr(), ws_ptr, C.c_void_p(stream)), "eec_ctc_loss_forward")
        ctx.save_for_backward(enc_out, tg, tl, nll, ws)
        ctx.blank = blank
        return out

    @staticmethod
    def backward(ctx, grad_out):
        enc_out, tg, tl, nll, ws = ctx.saved_tensors
        if getattr(ctx, "used", False):
            raise RuntimeError("exit_ctc_losses: backward through the same forward twice (its workspace is consumed)")
        ctx.used = True
        E, B, Tq, V = enc_out.shape
        dev = enc_out.device
        g = grad_out.to(device=dev, dtype=torch.float32).contiguous()
        dlogp = torch.empty_like(enc_out)
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            capi.check(capi.load().eec_ctc_loss_backward(enc_out.data_ptr(), tg.data_ptr(), tl.data_ptr(), E, B, Tq, V, tg.size(1),
                                                         ctx.blank, nll.data_ptr(), ws_ptr, g.data_ptr(), dlogp.data_ptr(),
                                                         C.c_void_p(stream)), "eec_ctc_loss_backward")
        return dlogp, None, None, None


def exit_ctc_losses(enc_out: Tensor, targets: Tensor, target_len: Tensor, blank: int = 0) -> Tensor:
    """Per-exit CTC losses [E] of an encoder output [E, B, T', V] in ONE launch: what train.py:53-65 computes with
    E separate nn.CTCLoss(blank=0, reduction='mean', zero_infinity=True) calls and input length T' for every
    utterance.  ``.sum()`` is the reference's training loss.  Differentiable with respect to ``enc_out`` (HIP backward:
    beta recursion + dense gradient, the values torch autograd returns for the reference's loop)."""
    enc_out, tg, tl = _ctc_prepare(enc_out, targets, target_len)
    E, B, Tq, V = enc_out.shape
    if torch.is_grad_enabled() and enc_out.requires_grad:
        return _ExitCtcLossFn.apply(enc_out, tg, tl, blank)
    dev = enc_out.device
    nll = torch.empty((E * B,), dtype=torch.float32, device=dev)
    out = torch.empty((E,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        capi.check(capi.load().eec_ctc_loss(enc_out.data_ptr(), tg.data_ptr(), tl.data_ptr(), E, B, Tq, V, tg.size(1), blank,
                                            nll.data_ptr(), out.data_ptr(), C.c_void_p(stream)), "eec_ctc_loss")
    return out


class _ExitHeadsFn(torch.autograd.Function):
    """All exit heads on given encoder taps: log_softmax(taps[e] . W_e^T + b_e) (early_exit.py:629-631), forward through
    the HIP head kernel, backward = HIP log-softmax backward + the two GEMMs of a Linear's backward on the training GEMM
    (eec_train_head_backward)."""

    @staticmethod
    def forward(ctx, model, taps, *wb):
        E, B, Tq, D = taps.shape
        V = model._cfg.vocab
        dev = taps.device
        lib = capi.load()
        out = torch.empty((E, B, Tq, V), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            for e in range(E):
                capi.check(lib.eec_encoder_head_forward(model._enc, e, taps[e].data_ptr(), B * Tq, out[e].data_ptr(),
                                                        capi.PRECISIONS[model.precision], C.c_void_p(stream)), "eec_encoder_head_forward")
        ctx.save_for_backward(taps, out, *wb[:E])
        ctx.need_taps = taps.requires_grad
        return out

    @staticmethod
    def backward(ctx, g):
        taps, out = ctx.saved_tensors[:2]
        ws = ctx.saved_tensors[2:]
        E, B, Tq, D = taps.shape
        V = out.size(-1)
        dev = taps.device
        g = g.contiguous().float()
        lib = capi.load()
        M = B * Tq
        dW = [torch.empty_like(w) for w in ws]
        db = [torch.empty((V,), dtype=torch.float32, device=dev) for _ in range(E)]
        dtaps = torch.empty_like(taps) if ctx.need_taps else None
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            scratch = torch.empty((lib.eec_train_head_backward_scratch_floats(M, V, D),), dtype=torch.float32, device=dev)
            for e in range(E):  # log-softmax backward + the two GEMMs of a Linear's backward on the training GEMM (bf16x3)
                _trainer_check(lib.eec_train_head_backward(taps[e].data_ptr(), ws[e].data_ptr(), out[e].data_ptr(), g[e].data_ptr(), M, V, D, 3,
                                                           dtaps[e].data_ptr() if dtaps is not None else None, dW[e].data_ptr(), db[e].data_ptr(),
                                                           scratch.data_ptr(), C.c_void_p(stream)), "eec_train_head_backward")
            scratch.record_stream(torch.cuda.current_stream(dev))
        return (None, dtaps, *dW, *db)


def _trainer_check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {capi.load().eec_trainer_last_error().decode(errors='replace')}")


def _params_struct(model, tensors: Dict[str, Optional[Tensor]]):
    """EecParams over ``tensors`` (state_dict names -> tensor or None); returns (struct, keep-alive list)."""
    E, L = model._cfg.n_exits, model._cfg.layers_per_exit

    def ptr(name: str):
        t = tensors.get(name)
        return t.data_ptr() if t is not None else None

    layers = _layer_params(ptr, "conformer", E, L)
    hw = (C.c_void_p * E)(*[ptr(f"{model._head_attr}.{e}.weight") for e in range(E)])
    hb = (C.c_void_p * E)(*[ptr(f"{model._head_attr}.{e}.bias") for e in range(E)])
    st = capi.EecParams(ptr("conv_subsample.sequential.0.weight"), ptr("conv_subsample.sequential.0.bias"),
                        ptr("conv_subsample.sequential.1.weight"), ptr("conv_subsample.sequential.1.bias"),
                        ptr(f"{model._pe_attr}.pe"), layers, hw, hb)
    return st, (layers, hw, hb)


def _named_tensors(model):
    """(named parameters, state_dict entries) of ``model`` as lists of (name, tensor) -- the names and order of
    ``named_parameters()`` / ``state_dict(keep_vars=True)`` -- from an index of (name, module, key) built once per model: the
    training step asks twice per forward and the walk over ~400 modules was 1.7 ms of its host time.  The tensors are read from the
    modules at every call (``.to()``, ``load_state_dict`` and in-place updates are seen); modules added later are not."""
    idx = model.__dict__.get("_tensor_index")
    if idx is None:
        pidx, sidx, seen = [], [], set()
        for mname, mod in model.named_modules(remove_duplicate=False):  # state_dict() lists a shared module under every path
            pre = mname + "." if mname else ""
            for k, v in mod._parameters.items():
                if v is not None:
                    sidx.append((pre + k, mod, k, True))
                    if id(v) not in seen:  # named_parameters() lists a shared parameter once
                        seen.add(id(v))
                        pidx.append((pre + k, mod, k))
            for k, v in mod._buffers.items():
                if v is not None and k not in mod._non_persistent_buffers_set:
                    sidx.append((pre + k, mod, k, False))
        idx = model.__dict__["_tensor_index"] = (pidx, sidx)
    pidx, sidx = idx
    return ([(n, m._parameters[k]) for n, m, k in pidx],
            [(n, (m._parameters if is_p else m._buffers)[k]) for n, m, k, is_p in sidx])


class _EncoderTrainFn(torch.autograd.Function):
    """``Early_conformer.forward`` in train mode and its backward on the HIP training kernels (csrc/train.hip): what
    ``enc_out = model(batch_0, valid_lengths)`` / ``loss.backward()`` do in the reference's train.py:53-68.  BatchNorm uses
    the batch statistics (and updates running_mean / running_var / num_batches_tracked like nn.BatchNorm1d), dropout
    runs at the reference's sites with probability ``model.dropout``."""

    @staticmethod
    def forward(ctx, model, src, len_dev, names, want_taps, *params):
        lib = capi.load()
        dev = src.device
        cfg = model._cfg
        B, _, T = src.shape
        Tq = lib.eec_out_frames(T)
        E, L, D, V = cfg.n_exits, cfg.layers_per_exit, cfg.d_model, cfg.vocab
        with torch.cuda.device(dev):
            if getattr(model, "_trainer", None) is None or model._trainer_device != dev:
                if getattr(model, "_trainer", None) is not None:
                    lib.eec_trainer_destroy(model._trainer)
                h = C.c_void_p()
                _trainer_check(lib.eec_trainer_create(C.byref(cfg), C.byref(h)), "eec_trainer_create")
                model._trainer, model._trainer_device = h, dev
            tensors = dict(zip(names, params))
            for k, v in _named_tensors(model)[1]:  # what model.state_dict(keep_vars=True) holds, without walking the module tree again
                tensors.setdefault(k, v)
            for k, t in tensors.items():
                if t.is_floating_point() and (t.device != dev or t.dtype != torch.float32 or not t.is_contiguous()):
                    raise RuntimeError(f"parameter {k} must be a contiguous fp32 tensor on {dev}")
            pst, keep = _params_struct(model, tensors)
            nbytes = lib.eec_trainer_workspace_bytes(model._trainer, B, T)
            if nbytes == 0:
                raise ValueError("unsupported geometry for the training step")
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
            ws_ptr = (ws.data_ptr() + 255) // 256 * 256
            out = torch.empty((E, B, Tq, V), dtype=torch.float32, device=dev)
            taps = torch.empty((E, B, Tq, D), dtype=torch.float32, device=dev) if want_taps else None
            bn = torch.empty((E * L, 2, D), dtype=torch.float32, device=dev)
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            stream = torch.cuda.current_stream(dev).cuda_stream
            _trainer_check(lib.eec_train_forward(model._trainer, C.byref(pst), src.data_ptr(), len_dev.data_ptr(), B, T,
                                                 int(model.train_passes), float(model.dropout), seed, out.data_ptr(),
                                                 taps.data_ptr() if want_taps else None, bn.data_ptr(), ws_ptr, nbytes, C.c_void_p(stream)),
                           "eec_train_forward")
            model._train_generation = getattr(model, "_train_generation", 0) + 1
            ctx.generation = model._train_generation
            # running statistics, as nn.BatchNorm1d(momentum=0.1) in train mode
            n = B * Tq
            li = 0
            with torch.no_grad():
                # one multi-tensor update per momentum value (12 layers x 6 tiny kernels otherwise): lists of (layer index, module)
                by_m = {}
                for grp in model.conformer:
                    for layer in grp.conformer_layers:
                        bnm = layer.conv_module.sequential[3]
                        if bnm.track_running_stats and bnm.running_mean is not None:
                            by_m.setdefault(bnm.momentum if bnm.momentum is not None else 0.1, []).append((li, bnm))
                        li += 1
                if by_m:
                    bvar = bn[:, 1] * (n / max(n - 1, 1))  # unbiased, as nn.BatchNorm1d stores it
                    for m, mods in by_m.items():
                        means, vars_ = [b_.running_mean for _, b_ in mods], [b_.running_var for _, b_ in mods]
                        torch._foreach_mul_(means, 1 - m)
                        torch._foreach_add_(means, [bn[i, 0] for i, _ in mods], alpha=m)
                        torch._foreach_mul_(vars_, 1 - m)
                        torch._foreach_add_(vars_, [bvar[i] for i, _ in mods], alpha=m)
                        torch._foreach_add_([b_.num_batches_tracked for _, b_ in mods], 1)
        ctx.model, ctx.names, ctx.ws, ctx.ws_ptr, ctx.nbytes = model, names, ws, ws_ptr, nbytes
        ctx.keep = (src, len_dev)
        ctx.want_taps = bool(want_taps)
        ctx.save_for_backward(out, *params)
        return (out, taps) if want_taps else out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, g_taps=None):
        model, names = ctx.model, ctx.names
        if ctx.generation != model._train_generation:
            raise RuntimeError("the trainer records one forward at a time: run backward before the next training forward")
        out, params = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        dev = out.device
        lib = capi.load()
        g = torch.zeros_like(out) if g is None else g.contiguous().float()
        g_taps = g_taps.contiguous().float() if g_taps is not None else None
        with torch.cuda.device(dev):
            tensors = dict(zip(names, params))
            for k, v in model.state_dict(keep_vars=True).items():
                tensors.setdefault(k, v)
            pst, keep = _params_struct(model, tensors)
            # data-parallel mode (enable_data_parallel): gradients are written straight into the flat buckets and each exit
            # group's bucket is all-reduced as soon as the backward has passed that group.  Only while no parameter holds a
            # gradient yet (zero_grad(set_to_none=True), the torch default): autograd then installs the views as p.grad;
            # otherwise it would ADD the view to a p.grad that may alias it, so the step falls back to fresh tensors and
            # sync_gradients() reduces afterwards.
            dp = getattr(model, "_dp", None)
            use_views = dp is not None and all(p.grad is None for p in params)
            grads = {}
            for k, v in zip(names, params):
                view = dp["buckets"].view(k, v) if use_views else None
                grads[k] = view if view is not None else torch.empty_like(v)
            gst, gkeep = _params_struct(model, grads)
            stream = torch.cuda.current_stream(dev).cuda_stream
            cb, err = capi.GROUP_DONE_FN(0), []
            if dp is not None and dp["active"]:
                buckets, weight, group = dp["buckets"], dp["weight"], dp["group"]
                # a bucket may also hold parameters this function does not differentiate (full_conformer's decoders: their
                # gradients are autograd's own tensors): such buckets, and every bucket when the views are not in use, are
                # left to sync_gradients()
                mine = set(names)
                early = [use_views and all(n in mine for n, _ in b["params"]) for b in buckets.buckets]
                dp["reduced"] = set()

                def on_group(e, _user):
                    try:
                        for i in buckets.buckets_ready_after(e):
                            if early[i]:
                                buckets.allreduce_bucket(i, weight, group, trusted=True)
                                dp["reduced"].add(i)
                    except Exception as ex:  # never unwind through the C frames
                        err.append(ex)
                if any(early):
                    cb = capi.GROUP_DONE_FN(on_group)
            _trainer_check(lib.eec_train_backward_ex(model._trainer, C.byref(pst), C.byref(gst), out.data_ptr(), g.data_ptr(),
                                                     g_taps.data_ptr() if g_taps is not None else None, ctx.ws_ptr, ctx.nbytes,
                                                     C.c_void_p(stream), cb, None), "eec_train_backward")
            if err:
                raise err[0]
        ctx.ws = None
        need = ctx.needs_input_grad[5:]
        return (None, None, None, None, None, *[grads[k] if nd else None for k, nd in zip(names, need)])


def beam_select(logp: Tensor, scores: Tensor, penalty: float, k: int, tokens_old: Tensor, tokens_new: Tensor, length: int):
    """One step of beam-search bookkeeping for n searches in lockstep, in one launch (eec_beam_select): the ``k`` best of
    ``scores[i, r] + logp[i, r, v] / penalty`` per search i, best first -> ``(scores [n, k], parent [n, k], token [n, k])``,
    and ``tokens_new[i, b, :length + 1] = cat(tokens_old[i, parent[i, b], :length], token[i, b])``.  What
    util/beam_infer.py:241-262 does with topk / index / cat, for every exit of an utterance at once."""
    n, R, V = logp.shape
    dev = logp.device
    if tokens_old.shape != tokens_new.shape or tokens_old.dim() != 3 or tokens_old.size(0) != n:
        raise ValueError("token buffers: two [n, rows, steps] int64 tensors")
    out_s = torch.empty((n, k), dtype=torch.float32, device=dev)
    parent = torch.empty((n, k), dtype=torch.int64, device=dev)
    tok = torch.empty((n, k), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        lib = capi.load()
        rc = lib.eec_beam_select(n, R, V, int(k), logp.contiguous().data_ptr(), scores.contiguous().data_ptr(), float(penalty), out_s.data_ptr(),
                                 parent.data_ptr(), tok.data_ptr(), tokens_old.data_ptr(), tokens_new.data_ptr(), int(length),
                                 tokens_old.size(2), tokens_old.size(1), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"eec_beam_select failed (code {rc}): {lib.eec_decoder_step_last_error().decode(errors='replace')}")
    return out_s, parent, tok


class DecoderSession:
    """One utterance's step-wise AED decoding state (include/eec.h, eec_decoder_begin / eec_decoder_step): ``step(tokens,
    parent)`` returns the log-probs of the NEXT token for every live beam, [R, V] -- what
    ``model._decoder_(prefixes, enc, layer_n)[:, -1]`` returns (util/beam_infer.py:236-240) -- from the last token of every
    beam and the row of the previous step it extends."""

    def __init__(self, model, ps, d_ff: int, V: int, enc: Tensor, max_steps: int, nbytes: int):
        lib = capi.load()
        self.model, self.ps, self.d_ff, self.V, self.max_steps, self.nbytes = model, ps, d_ff, V, max_steps, nbytes
        self.dev, self.Tq = enc.device, enc.size(0)
        self.s, self.rows = 0, 0
        self.max_beams = lib.eec_decoder_step_max_beams()
        cfg = model._cfg
        with torch.cuda.device(self.dev):
            self.cache = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.dev)
            self.ptr = (self.cache.data_ptr() + 255) // 256 * 256
            enc_c = enc.contiguous().float()
            stream = torch.cuda.current_stream(self.dev)
            rc = lib.eec_decoder_begin(C.byref(ps), cfg.d_model, cfg.n_heads, d_ff, V, enc_c.data_ptr(), self.Tq, max_steps,
                                       int(model.decoder_passes), self.ptr, nbytes, C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError(f"eec_decoder_begin failed (code {rc}): {lib.eec_decoder_step_last_error().decode(errors='replace')}")
            enc_c.record_stream(stream)
            self.cache.record_stream(stream)

    def step(self, last_tokens: Tensor, parent: Optional[Tensor] = None, log_softmax: bool = True) -> Tensor:
        lib = capi.load()
        cfg = self.model._cfg
        R = int(last_tokens.numel())
        if not 1 <= R <= self.max_beams:
            raise ValueError(f"1 .. {self.max_beams} live beams per step, got {R}")
        if self.s >= self.max_steps:
            raise RuntimeError(f"the session was opened for {self.max_steps} steps")
        if parent is not None and parent.numel() != R:
            raise ValueError("parent: one row of the previous step per live beam")
        with torch.cuda.device(self.dev):
            tok = last_tokens.to(device=self.dev, dtype=torch.int64).contiguous()
            par = parent.to(device=self.dev, dtype=torch.int64).contiguous() if parent is not None and self.s > 0 else None
            out = torch.empty((R, self.V), dtype=torch.float32, device=self.dev)
            stream = torch.cuda.current_stream(self.dev)
            rc = lib.eec_decoder_step(C.byref(self.ps), cfg.d_model, cfg.n_heads, self.d_ff, self.V, int(self.model.trg_pad_idx),
                                      tok.data_ptr(), par.data_ptr() if par is not None else None, R, self.rows, self.s, self.Tq,
                                      self.max_steps, int(log_softmax), out.data_ptr(), self.ptr, self.nbytes,
                                      C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError(f"eec_decoder_step failed (code {rc}): {lib.eec_decoder_step_last_error().decode(errors='replace')}")
            tok.record_stream(stream)
            if par is not None:
                par.record_stream(stream)
        self.s += 1
        self.rows = R
        return out


class DecoderSessionGroup:
    """The step-wise decoding sessions of several exits of ONE utterance advanced in lockstep by the same launches
    (eec_decoder_step_multi): ``step(tokens [n, R], parent [n, R])`` -> log-probs [n, R, V]."""

    MAX = 8

    def __init__(self, sessions: List["DecoderSession"]):
        s0 = sessions[0]
        if not 1 <= len(sessions) <= self.MAX:
            raise ValueError(f"1 .. {self.MAX} sessions per group")
        for s in sessions:
            if (s.dev, s.Tq, s.max_steps, s.nbytes, s.V, s.d_ff, s.s) != (s0.dev, s0.Tq, s0.max_steps, s0.nbytes, s0.V, s0.d_ff, 0):
                raise ValueError("the sessions of a group share device, geometry and step budget, and have not stepped yet")
        self.sessions = sessions
        n = len(sessions)
        self.ps = (C.POINTER(capi.EecDecoderParams) * n)(*[C.pointer(s.ps) for s in sessions])
        self.caches = (C.c_void_p * n)(*[s.ptr for s in sessions])
        self.s, self.rows, self.max_beams = 0, 0, s0.max_beams

    def step(self, last_tokens: Tensor, parent: Optional[Tensor] = None, log_softmax: bool = True) -> Tensor:
        lib = capi.load()
        s0 = self.sessions[0]
        cfg = s0.model._cfg
        n = len(self.sessions)
        if last_tokens.dim() != 2 or last_tokens.size(0) != n:
            raise ValueError(f"last_tokens must be [{n}, live beams]")
        R = int(last_tokens.size(1))
        if not 1 <= R <= self.max_beams:
            raise ValueError(f"1 .. {self.max_beams} live beams per step, got {R}")
        if self.s >= s0.max_steps:
            raise RuntimeError(f"the sessions were opened for {s0.max_steps} steps")
        if parent is not None and tuple(parent.shape) != (n, R):
            raise ValueError("parent: one row of the previous step per live beam and session")
        with torch.cuda.device(s0.dev):
            tok = last_tokens.to(device=s0.dev, dtype=torch.int64).contiguous()
            par = parent.to(device=s0.dev, dtype=torch.int64).contiguous() if parent is not None and self.s > 0 else None
            out = torch.empty((n, R, s0.V), dtype=torch.float32, device=s0.dev)
            stream = torch.cuda.current_stream(s0.dev)
            rc = lib.eec_decoder_step_multi(n, self.ps, cfg.d_model, cfg.n_heads, s0.d_ff, s0.V, int(s0.model.trg_pad_idx), tok.data_ptr(),
                                            par.data_ptr() if par is not None else None, R, self.rows, self.s, s0.Tq, s0.max_steps,
                                            int(log_softmax), out.data_ptr(), self.caches, s0.nbytes, C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError(f"eec_decoder_step_multi failed (code {rc}): {lib.eec_decoder_step_last_error().decode(errors='replace')}")
            tok.record_stream(stream)
            if par is not None:
                par.record_stream(stream)
        self.s += 1
        self.rows = R
        for s in self.sessions:  # a session that joined a group is advanced only through it
            s.s, s.rows = self.s, R
        return out


class _DecoderTrainFn(torch.autograd.Function):
    """Exit ``idx``'s attention decoder in train mode and its backward on the HIP training kernels (eec_decoder_train_forward /
    _backward): ``linears_2[idx](decoders[idx](positional_encoder_2(emb(trg)), enc, causal + padding masks))`` -> raw logits
    [B, S, V], differentiable with respect to every decoder parameter, the embedding table and ``enc`` (the encoder tap)."""

    @staticmethod
    def forward(ctx, model, idx, trg, enc, seed, names, *params):
        lib = capi.load()
        dev = trg.device
        cfg = model._cfg
        Bm, S = trg.shape
        Tq = enc.size(1)
        if enc.size(0) != Bm or enc.size(2) != cfg.d_model:
            raise ValueError(f"enc must be [{Bm}, T', {cfg.d_model}], got {tuple(enc.shape)}")
        tensors = dict(zip(names, params))
        for k, t in tensors.items():
            if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError(f"parameter {k} must be a contiguous fp32 tensor on {dev}")
        d_ff = model.decoders[idx].layers[0].linear1.out_features
        V = model.linears_2[idx].out_features
        n_layers = len(model.decoders[idx].layers)
        with torch.cuda.device(dev):
            ps, keep = model._decoder_struct(idx, tensors, with_pe=True)
            nbytes = lib.eec_decoder_train_workspace_bytes(cfg.d_model, cfg.n_heads, d_ff, V, n_layers, Bm, S, Tq)
            if nbytes == 0:
                raise ValueError("unsupported geometry for the decoder's training step")
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
            ws_ptr = (ws.data_ptr() + 255) // 256 * 256
            out = torch.empty((Bm, S, V), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            geo = (cfg.d_model, cfg.n_heads, d_ff, V)
            rc = lib.eec_decoder_train_forward(C.byref(ps), *geo, int(model.trg_pad_idx), trg.data_ptr(), enc.data_ptr(), Bm, S, Tq,
                                               int(model.decoder_passes), float(model.dropout), int(seed), int(idx), out.data_ptr(), ws_ptr, nbytes,
                                               C.c_void_p(stream))
            if rc != 0:
                raise RuntimeError(f"eec_decoder_train_forward failed (code {rc}): {lib.eec_decoder_train_last_error().decode(errors='replace')}")
        ctx.model, ctx.idx, ctx.names, ctx.seed, ctx.geo = model, idx, names, int(seed), geo
        ctx.ws, ctx.ws_ptr, ctx.nbytes, ctx.drop, ctx.passes = ws, ws_ptr, nbytes, float(model.dropout), int(model.decoder_passes)
        ctx.save_for_backward(trg, enc, *params)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if ctx.ws is None:
            raise RuntimeError("the decoder's recorded forward was already consumed by a backward")
        model, idx, names = ctx.model, ctx.idx, ctx.names
        trg, enc, params = ctx.saved_tensors[0], ctx.saved_tensors[1], ctx.saved_tensors[2:]
        dev = trg.device
        lib = capi.load()
        Bm, S = trg.shape
        Tq = enc.size(1)
        g = g.contiguous().float()
        with torch.cuda.device(dev):
            tensors = dict(zip(names, params))
            ps, keep = model._decoder_struct(idx, tensors, with_pe=True)
            grads = {k: torch.empty_like(v) for k, v in tensors.items()}
            gs, gkeep = model._decoder_struct(idx, grads, with_pe=False)
            g_enc = torch.empty_like(enc)
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = lib.eec_decoder_train_backward(C.byref(ps), C.byref(gs), *ctx.geo, trg.data_ptr(), enc.data_ptr(), Bm, S, Tq, ctx.passes, ctx.drop,
                                                ctx.seed, int(idx), g.data_ptr(), g_enc.data_ptr(), ctx.ws_ptr, ctx.nbytes, C.c_void_p(stream))
            if rc != 0:
                raise RuntimeError(f"eec_decoder_train_backward failed (code {rc}): {lib.eec_decoder_train_last_error().decode(errors='replace')}")
        ctx.ws = None
        need = ctx.needs_input_grad[6:]
        return (None, None, None, g_enc if ctx.needs_input_grad[3] else None, None, None,
                *[grads[k] if nd else None for k, nd in zip(names, need)])


class full_conformer(_HipEncoderMixin, nn.Module):
    """AED model: HIP encoder + the attention decoder, both on the hand-written path: inference through csrc/decoder.hip /
    decoder_step.hip, training (forward in train mode + backward, train.py:36-52) through csrc/decoder_train.hip
    behind an autograd function.  The ``nn.TransformerDecoder`` modules only hold the parameters (state_dict contract)."""

    _head_attr = "linears_1"
    _pe_attr = "positional_encoder_1"

    def __init__(self, trg_pad_idx, n_enc_exits, enc_voc_size, dec_voc_size, d_model, n_head, max_len,
                 d_feed_forward, n_enc_layers, n_dec_layers, features_length, drop_prob, depthwise_kernel_size,
                 device=None):
        nn.Module.__init__(self)
        self.input_dim, self.num_heads, self.ffn_dim = d_model, n_head, d_feed_forward
        self.num_layers, self.depthwise_conv_kernel_size = n_enc_layers, depthwise_kernel_size
        self.n_enc_exits, self.dropout, self.n_dec_layers = n_enc_exits, drop_prob, n_dec_layers
        self.device, self.trg_pad_idx = device, trg_pad_idx
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-5)
        self.emb = nn.Embedding(dec_voc_size, d_model)
        self.conv_subsample = Conv1dSubampling(features_length, d_model)
        self.linears_1 = nn.ModuleList([nn.Linear(d_model, dec_voc_size) for _ in range(n_enc_exits)])
        self.linears_2 = nn.ModuleList([nn.Linear(d_model, dec_voc_size) for _ in range(n_enc_exits)])
        self.positional_encoder_1 = PositionalEncoding(d_model, drop_prob, max_len)
        self.positional_encoder_2 = PositionalEncoding(d_model, drop_prob, max_len)
        self.conformer = nn.ModuleList([
            Conformer(input_dim=d_model, num_heads=n_head, ffn_dim=d_feed_forward, num_layers=n_enc_layers,
                      depthwise_conv_kernel_size=depthwise_kernel_size, dropout=drop_prob)
            for _ in range(n_enc_exits)])
        self.decoders = nn.ModuleList([
            nn.TransformerDecoder(
                nn.TransformerDecoderLayer(d_model=d_model, nhead=n_head, dim_feedforward=d_feed_forward,
                                           dropout=drop_prob, batch_first=True, norm_first=True),
                n_dec_layers, self.layer_norm)
            for _ in range(n_enc_exits)])
        self._hip_init(d_model, n_head, d_feed_forward, depthwise_kernel_size, n_enc_exits, n_enc_layers,
                       features_length, dec_voc_size, max_len)

    def _encoder_(self, src: Tensor, lengths: Tensor, layer_n: int) -> Tensor:
        """Pre-head activations after ``layer_n`` exit groups, [B, T', D] (early_exit.py:719-737)."""
        n = int(layer_n) if 1 <= int(layer_n) <= self._cfg.n_exits else self._cfg.n_exits  # reference loop
        return self._run_encoder(src, lengths, want_out=False, want_x=True, n_groups=n)[2]

    decoder_passes = 3  # the HIP decoder's GEMM operands: 3 = bf16 hi/lo split (~1e-5 of fp32), 1 = plain bf16

    def _decode_one(self, trg: Tensor, enc: Tensor, idx: int, log_softmax: bool = False, seed: Optional[int] = None) -> Tensor:
        if not trg.is_cuda:
            raise RuntimeError("the MI355X decoder runs on a HIP device only (there is no CPU fallback)")
        if not self.training:
            return self._hip_decoder(trg, enc, idx, log_softmax)  # inference: csrc/decoder.hip
        # train mode (train.py:36-52), with or without autograd -- the reference's modules apply their dropout in train mode
        # whatever the grad mode: forward (and backward) on the HIP training kernels (csrc/decoder_train.hip)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        named = self._decoder_named_params(idx)
        out = _DecoderTrainFn.apply(self, idx, trg.to(torch.int64).contiguous(), enc.contiguous().float(), seed,
                                    tuple(n for n, _ in named), *[p for _, p in named])
        return torch.log_softmax(out, dim=2) if log_softmax else out

    def _decoder_named_params(self, idx: int):
        """(name, parameter) of everything exit ``idx``'s decoder reads, under the model's state_dict names (the decoders'
        final norm is the ONE shared ``layer_norm``, early_exit.py:666,701-717)."""
        named = [("emb.weight", self.emb.weight), ("layer_norm.weight", self.layer_norm.weight), ("layer_norm.bias", self.layer_norm.bias),
                 (f"linears_2.{idx}.weight", self.linears_2[idx].weight), (f"linears_2.{idx}.bias", self.linears_2[idx].bias)]
        for l, layer in enumerate(self.decoders[idx].layers):
            sd = dict(layer.named_parameters())
            named += [(f"decoders.{idx}.layers.{l}.{suffix}", sd[suffix]) for suffix in capi.DECODER_LAYER_KEYS.values()]
        return named

    def _decoder_struct(self, idx: int, tensors: Dict[str, Tensor], with_pe: bool):
        """eec_decoder_params over ``tensors`` (name -> tensor: the parameters themselves, or gradient buffers of their shapes)."""
        n_layers = len(self.decoders[idx].layers)
        layers = (capi.EecDecoderLayerParams * n_layers)()
        for l in range(n_layers):
            for field, suffix in capi.DECODER_LAYER_KEYS.items():
                setattr(layers[l], field, tensors[f"decoders.{idx}.layers.{l}.{suffix}"].data_ptr())
        pe = self.positional_encoder_2.pe
        ps = capi.EecDecoderParams(tensors["emb.weight"].data_ptr(), pe.data_ptr() if with_pe else None, layers, n_layers, pe.size(0),
                                   tensors["layer_norm.weight"].data_ptr(), tensors["layer_norm.bias"].data_ptr(),
                                   tensors[f"linears_2.{idx}.weight"].data_ptr(), tensors[f"linears_2.{idx}.bias"].data_ptr())
        return ps, layers

    def _decoder_params(self, idx: int, dev):
        """The C-ABI view (eec_decoder_params) of exit ``idx``'s decoder: pointers into the module's own parameters, rebuilt
        when any of them moved."""
        dec = self.decoders[idx]
        tensors = [self.emb.weight, self.positional_encoder_2.pe, self.layer_norm.weight, self.layer_norm.bias,
                   self.linears_2[idx].weight, self.linears_2[idx].bias] + list(dec.layers.parameters())
        key = (dev, tuple(t.data_ptr() for t in tensors))
        cache = self.__dict__.setdefault("_dec_cache", {})
        ent = cache.get(idx)
        if ent is None or ent[0] != key:
            for t in tensors:
                if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous():
                    raise RuntimeError(f"decoder parameters must be contiguous fp32 tensors on {dev}")
            layers = (capi.EecDecoderLayerParams * len(dec.layers))()
            for l, layer in enumerate(dec.layers):
                sd = dict(layer.named_parameters())
                for field, name in capi.DECODER_LAYER_KEYS.items():
                    setattr(layers[l], field, sd[name].data_ptr())
            ps = capi.EecDecoderParams(self.emb.weight.data_ptr(), self.positional_encoder_2.pe.data_ptr(), layers, len(dec.layers),
                                       self.positional_encoder_2.pe.size(0), self.layer_norm.weight.data_ptr(),
                                       self.layer_norm.bias.data_ptr(), self.linears_2[idx].weight.data_ptr(),
                                       self.linears_2[idx].bias.data_ptr())
            ent = (key, ps, layers)
            cache[idx] = ent
        return ent[1], dec.layers[0].linear1.out_features, self.linears_2[idx].out_features

    def _hip_decoder(self, trg: Tensor, enc: Tensor, idx: int, log_softmax: bool) -> Tensor:
        """``linears_2[idx](decoders[idx](positional_encoder_2(emb(trg)), enc, causal + padding masks))`` in eval semantics
        through eec_decoder_forward; trg int64 [Bm, S], enc fp32 [Bm, T', D] -> [Bm, S, V] logits or log-probs."""
        lib = capi.load()
        dev = trg.device
        cfg = self._cfg
        Bm, S = trg.shape
        Tq = enc.size(1)
        if enc.size(0) != Bm or enc.size(2) != cfg.d_model:
            raise ValueError(f"enc must be [{Bm}, T', {cfg.d_model}], got {tuple(enc.shape)}")
        ps, d_ff, V = self._decoder_params(idx, dev)
        with torch.cuda.device(dev):
            trg_c = trg.to(torch.int64).contiguous()
            shared = Bm > 1 and enc.stride(0) == 0  # beam search: one utterance expanded over the beams (util/beam_infer.py:233)
            enc_c = (enc[:1] if shared else enc).contiguous().float()
            nbytes = lib.eec_decoder_workspace_bytes(cfg.d_model, cfg.n_heads, d_ff, V, Bm, S, Tq)
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
            ws_ptr = (ws.data_ptr() + 255) // 256 * 256
            out = torch.empty((Bm, S, V), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = lib.eec_decoder_forward(C.byref(ps), cfg.d_model, cfg.n_heads, d_ff, V, int(self.trg_pad_idx), trg_c.data_ptr(),
                                         enc_c.data_ptr(), Bm, S, Tq, int(shared), int(self.decoder_passes), int(log_softmax), out.data_ptr(),
                                         ws_ptr, nbytes, C.c_void_p(stream))
            if rc != 0:
                raise RuntimeError(f"eec_decoder_forward failed (code {rc}): {lib.eec_decoder_last_error().decode(errors='replace')}")
            for t in (ws, trg_c, enc_c):
                t.record_stream(torch.cuda.current_stream(dev))
        return out

    def decoder_session(self, enc: Tensor, layer_n: int, max_steps: int) -> Optional["DecoderSession"]:
        """Step-wise decoding of ONE utterance with a key / value cache (csrc/decoder_step.hip): ``enc`` [1, T', D] or
        [T', D] is exit ``layer_n``'s encoder output, ``max_steps`` the longest prefix that will be decoded.  None when this
        geometry or device is not served (callers then use ``_decoder_`` on the whole prefix, as the reference does)."""
        if not enc.is_cuda or (self.training and torch.is_grad_enabled()):
            return None
        idx = (int(layer_n) if 1 <= int(layer_n) <= self._cfg.n_exits else self._cfg.n_exits) - 1
        enc2 = enc.reshape(-1, enc.size(-1)) if enc.dim() == 2 or enc.size(0) == 1 else None
        if enc2 is None or max_steps < 1 or max_steps > self.positional_encoder_2.pe.size(0):
            return None
        ps, d_ff, V = self._decoder_params(idx, enc.device)
        lib = capi.load()
        cfg = self._cfg
        nbytes = lib.eec_decoder_cache_bytes(cfg.d_model, cfg.n_heads, d_ff, V, len(self.decoders[idx].layers), int(max_steps), enc2.size(0))
        if nbytes == 0:
            return None
        return DecoderSession(self, ps, d_ff, V, enc2, int(max_steps), nbytes)

    def decoder_session_group(self, encs: Sequence[Tensor], layer_ns: Sequence[int], max_steps: int) -> Optional["DecoderSessionGroup"]:
        """Sessions for exits ``layer_ns`` of one utterance (``encs[i]``: that exit's encoder output), advanced together by
        ``group.step``; None when a session is not available or there are more than 8 exits."""
        if not 1 <= len(layer_ns) <= DecoderSessionGroup.MAX or len(encs) != len(layer_ns):
            return None
        sessions = [self.decoder_session(e, n, max_steps) for e, n in zip(encs, layer_ns)]
        if any(s is None for s in sessions):
            return None
        return DecoderSessionGroup(sessions)

    def _decoder_(self, trg: Tensor, enc: Tensor, layer_n: int) -> Tensor:
        idx = (int(layer_n) if 1 <= int(layer_n) <= self._cfg.n_exits else self._cfg.n_exits) - 1
        return self._decode_one(trg, enc, idx, log_softmax=True)

    def forward(self, src: Tensor, lengths: Tensor, trg: Tensor):
        if self.training:
            # train.py:36-52 (aed): encoder AND decoders forward / backward on the HIP training kernels (eec_train_*,
            # eec_decoder_train_*: _EncoderTrainFn, _DecoderTrainFn); the decoders consume the differentiable taps.  nn.TransformerDecoder
            # only holds the parameters.  Without autograd the same train-mode forward runs (dropout in both halves) and the tape is dropped
            enc_out, taps = Early_conformer._forward_train(self, src, lengths, want_taps=True)
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())  # one seed per forward: the exits share the embedding's dropout mask
            dec_out = torch.stack([self._decode_one(trg, taps[e], e, seed=seed) for e in range(self._cfg.n_exits)])
            return dec_out, enc_out
        enc_out, taps, _ = self._run_encoder(src, lengths, want_taps=True)
        dec_out = torch.stack([self._decode_one(trg, taps[e], e) for e in range(self._cfg.n_exits)])
        return dec_out, enc_out
