"""GPU tests of the training step (run with -m gpu on an MI355X): `Early_conformer` in train mode -- forward with
batch-statistics BatchNorm (+ dropout), summed per-exit CTC loss, loss.backward() -- on the HIP training kernels
(csrc/train.hip, train_kernels.hip) through the C ABI, against torch autograd on the CPU oracle (the reference's
train.py:53-68 arithmetic, fp32).

Tolerances: the training GEMMs split fp32 operands into bf16 hi + lo and sum three MFMA products (`train_passes = 3`):
every product is good to ~2^-16, so log-probs agree with the fp32 oracle to 2e-4 and every parameter gradient to 2e-3 of
that gradient's largest entry (measured values are printed).  `train_passes = 1` (plain bf16 operands, the north_star's
precision) is checked at 5e-2 / 0.15."""
import ctypes as C

import pytest
import torch

from conftest import base_kwargs, ref_decoder_logits
from early_exit_transformer_amd import capi, synth
from early_exit_transformer_amd.model import Early_conformer, Early_zipformer, Splitformer, exit_ctc_losses, full_conformer
from oracle import conformer_ref as R

pytestmark = pytest.mark.gpu

SMALL = dict(d_model=64, n_head=4, d_feed_forward=160, n_enc_exits=2, n_enc_layers=2, depthwise_kernel_size=7, dec_voc_size=32,
             enc_voc_size=32, max_len=200)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (37, 19, 5), (300, 70, 129), (64, 32, 256), (1, 1, 1), (200, 257, 33), (513, 96, 96)])
@pytest.mark.parametrize("at,bt", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_training_gemm(M, N, K, at, bt):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + at * 2 + bt)
    A, Bm, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    want = A.double() @ Bm.double().t() + bias.double()
    a_dev = (A.t().contiguous() if at else A).cuda()
    b_dev = (Bm.t().contiguous() if bt else Bm).cuda()
    lib = capi.load()
    for passes, tol in ((3, 3e-5), (1, 2e-2)):
        out = torch.full((M, N), float("nan"), device="cuda")
        rc = lib.eec_train_gemm(a_dev.data_ptr(), b_dev.data_ptr(), bias.cuda().data_ptr(), out.data_ptr(), M, N, K, passes, at, bt,
                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.eec_trainer_last_error()
        err = (out.cpu().double() - want).abs().max().item()
        assert err < tol * (K ** 0.5) * 4, (passes, err)


def make_train_pair(kw, seed, drop=0.0):
    kw = dict(kw, drop_prob=drop)
    ref = R.EarlyConformerRef(**kw)
    sd = synth.synth_state_dict(ref.state_dict(), seed=seed, style="trained")
    ref.load_state_dict(sd)
    gpu = Early_conformer(**{**kw, "device": "cuda"})
    gpu.load_state_dict(sd, strict=True)
    return ref.train(), gpu.cuda().train()


def grads_of(model):
    return {n: p.grad.detach().cpu().double() for n, p in model.named_parameters()}


def compare_grads(got, want, tol, label):
    worst = (0.0, "")
    gmax = max(g.abs().max().item() for g in want.values())
    for n, gw in want.items():
        # parameters whose exact gradient is zero (the depthwise bias in front of a batch-statistics BatchNorm) hold rounding
        # noise on both sides: the floor is relative to the largest gradient of the model
        scale = gw.abs().max().item() + 2e-3 * gmax
        err = (got[n] - gw).abs().max().item()
        rel = err / scale
        if rel > worst[0]:
            worst = (rel, n)
        assert torch.isfinite(got[n]).all(), n
        assert rel < tol, f"{label}: {n}: err {err:.3e} vs max|grad| {scale:.3e} (rel {rel:.2e})"
    print(f"\n[{label}] worst gradient error relative to that gradient's largest entry: {worst[0]:.2e} ({worst[1]})")


@pytest.mark.parametrize("cfg,B,T,lens", [
    (SMALL, 3, 131, [131, 90, 57]),
    (SMALL, 2, 151, [151, 100]),  # T' = 37: odd frame count -> the GEMMs' element-wise edge paths
    (dict(SMALL, n_head=2, depthwise_kernel_size=31, d_feed_forward=96), 5, 79, [79, 79, 60, 41, 30]),  # head dim 32, T' = 19
    (dict(n_enc_exits=2, n_enc_layers=1, d_feed_forward=512), 2, 259, [259, 170]),
    (dict(d_model=512, n_head=8, n_enc_exits=1, n_enc_layers=2, d_feed_forward=256, depthwise_kernel_size=31), 2, 99, [99, 64]),
    (dict(SMALL, n_head=2), 2, 47, [47, 30]),  # T' = 11: fused attention with a single, ragged key tile
    (dict(n_enc_exits=1, n_enc_layers=1, d_feed_forward=64), 1, 51, [51]),  # a single utterance: BatchNorm over its 12 frames
    # T' = 100 with 31 taps: the depthwise kernels' first / interior blocks take the compile-time forms, the last full block and the
    # 4-row remainder the generic one
    (dict(SMALL, n_head=2, depthwise_kernel_size=31), 2, 403, [403, 250]),
    (dict(SMALL, n_head=2, depthwise_kernel_size=31), 1, 259, [259]),  # T' = 64: a first and a last block, nothing between
    # the benchmark's frame count (T' = 256: two query / key blocks of the fused attention per head, eight key tiles, ragged keys)
    (dict(n_enc_exits=1, n_enc_layers=1, d_feed_forward=128), 2, 1027, [1027, 700]),
    (dict(d_model=512, n_head=8, n_enc_exits=1, n_enc_layers=1, d_feed_forward=128), 2, 1027, [1027, 513]),
    # the fused feed-forward launches (d_model 256 / 512, csrc/ffn.hip TR variants) at their edges: d_ff below one 128-wide chunk (three
    # 32-wide tiles: one producer wave idles), a 32-wide rest after full chunks, the reference's d_ff, row counts that are no multiple of
    # the 64- / 32-row tile (M = 3 x 32 = 96 and 2 x 37 = 74), a single short utterance (M = 12)
    (dict(n_enc_exits=1, n_enc_layers=1, d_feed_forward=96), 3, 131, [131, 100, 57]),
    (dict(n_enc_exits=1, n_enc_layers=1, d_feed_forward=416), 2, 151, [151, 90]),
    (dict(n_enc_exits=1, n_enc_layers=1, d_feed_forward=2048), 1, 51, [51]),
    (dict(d_model=512, n_head=8, n_enc_exits=1, n_enc_layers=1, d_feed_forward=160), 2, 151, [151, 120]),
])
def test_training_step_matches_oracle_autograd(cfg, B, T, lens):
    kw = base_kwargs(**cfg)
    ref, gpu = make_train_pair(kw, seed=31)
    mel, lens = synth.synth_mel(B, 80, T, seed=31), torch.tensor(lens)
    tgt, tl = synth.synth_targets(B, 9, kw["dec_voc_size"], seed=31)
    want_out = ref(mel, lens)
    want_loss = R.summed_exit_ctc_loss(want_out, tgt, tl)
    want_loss.backward()
    out = gpu(mel.cuda(), lens)
    assert out.requires_grad and out.shape == want_out.shape
    err = (out.detach().cpu() - want_out.detach()).abs().max().item()
    print(f"\n[train fwd] max |dlogp| vs the oracle in train mode: {err:.2e}")
    assert err < 2e-4
    loss = exit_ctc_losses(out, tgt, tl).sum()
    assert abs(loss.item() - want_loss.item()) < 2e-4 * max(1.0, abs(want_loss.item()))
    loss.backward()
    compare_grads(grads_of(gpu), grads_of(ref), 2e-3, "bf16x3")
    # BatchNorm running statistics were updated like nn.BatchNorm1d does in train mode
    for (n, b_ref), (_, b_gpu) in zip(ref.named_buffers(), gpu.named_buffers()):
        if "running_" in n or "num_batches" in n:
            assert torch.allclose(b_gpu.cpu().float(), b_ref.float(), rtol=1e-4, atol=1e-6), n
    # plain bf16 operands
    gpu.zero_grad()
    gpu.train_passes = 1
    out1 = gpu(mel.cuda(), lens)
    assert (out1.detach().cpu() - want_out.detach()).abs().max().item() < 5e-2
    exit_ctc_losses(out1, tgt, tl).sum().backward()
    compare_grads(grads_of(gpu), grads_of(ref), 0.15, "bf16")


@pytest.mark.parametrize("cfg,B,T,lens,tol", [
    # BASELINE.json configs[3]'s model in full depth: the default 12-layer d_model 256 network (6 exits x 2), T' = 256
    (dict(), 4, 1027, [1027, 903, 771, 642], 3e-3),
    # configs[2]'s geometry in full depth: 18 layers (6 exits x 3) at d_model 512, head dim 64; d_ff reduced for the CPU oracle's time
    (dict(d_model=512, n_enc_layers=3, d_feed_forward=512), 2, 515, [515, 400], 3e-3),
], ids=["default_12_layers", "18_layers_d512"])
def test_deep_training_step_matches_oracle_autograd(cfg, B, T, lens, tol):
    """Depth is what the shallow cases above do not cover: the bf16x3 rounding of every GEMM accumulates through 12 / 18 layers
    of forward and backward.  Same comparison (train.py:53-68: train-mode forward, summed exit CTC losses, backward; dropout 0
    so that the oracle's autograd is comparable), every one of the 413 / 605 parameter gradients."""
    kw = base_kwargs(**cfg)
    ref, gpu = make_train_pair(kw, seed=41)
    mel, lens = synth.synth_mel(B, 80, T, seed=41), torch.tensor(lens)
    tgt, tl = synth.synth_targets(B, 20, kw["dec_voc_size"], seed=41)
    want_out = ref(mel, lens)
    want_loss = R.summed_exit_ctc_loss(want_out, tgt, tl)
    want_loss.backward()
    out = gpu(mel.cuda(), lens)
    err = (out.detach().cpu() - want_out.detach()).abs().max().item()
    print(f"\n[deep train fwd] {kw['n_enc_exits'] * kw['n_enc_layers']} layers, d_model {kw['d_model']}: max |dlogp| vs the oracle in train mode: {err:.2e}")
    assert err < 5e-4
    loss = exit_ctc_losses(out, tgt, tl).sum()
    assert abs(loss.item() - want_loss.item()) < 3e-4 * max(1.0, abs(want_loss.item()))
    loss.backward()
    compare_grads(grads_of(gpu), grads_of(ref), tol, f"bf16x3, {kw['n_enc_exits'] * kw['n_enc_layers']} layers")


@pytest.mark.parametrize("which,cfg,B,T,lens", [
    ("splitformer", dict(SMALL, n_enc_exits=3, n_enc_layers=1), 3, 131, [131, 90, 57]),        # T' = 32 (even)
    ("splitformer", dict(SMALL, n_enc_exits=2, n_enc_layers=2, n_head=2), 2, 151, [151, 100]),  # T' = 37 (odd: the padded branch)
    ("zipformer", dict(SMALL, n_enc_exits=19, n_enc_layers=1, d_feed_forward=96), 2, 139, [139, 80]),  # T1 = 69: every stack pads
])
def test_other_model_types_train_on_the_hip_path(which, cfg, B, T, lens):
    """train.py:180-208: Splitformer / Early_zipformer in train mode -- stem, every Conformer group (main and down-sampled
    branches, five frame rates), heads on the HIP training kernels, the glue between them under torch autograd -- against torch
    autograd on the oracle's restatements of those classes (bit-identical to the reference's, tests/test_oracle.py), dropout 0:
    log-probs, loss, the gradient of every parameter, BatchNorm running statistics."""
    kw = base_kwargs(**dict(cfg, drop_prob=0.0))
    ref = (R.SplitformerRef if which == "splitformer" else R.EarlyZipformerRef)(**kw)
    sd = synth.synth_state_dict(ref.state_dict(), seed=53, style="trained")
    ref.load_state_dict(sd)
    gpu = (Splitformer if which == "splitformer" else Early_zipformer)(**{**kw, "device": "cuda"})
    gpu.load_state_dict(sd, strict=True)
    ref, gpu = ref.train(), gpu.cuda().train()
    mel, lens = synth.synth_mel(B, 80, T, seed=53), torch.tensor(lens)
    tgt, tl = synth.synth_targets(B, 5, kw["dec_voc_size"], seed=53)
    want_out = ref(mel, lens)
    want_loss = R.summed_exit_ctc_loss(want_out, tgt, tl)
    want_loss.backward()
    out = gpu(mel.cuda(), lens)
    assert out.requires_grad and out.shape == want_out.shape
    scale = max(1.0, want_out.detach().abs().max().item() / 8.0)
    err = (out.detach().cpu() - want_out.detach()).abs().max().item()
    print(f"\n[{which} train fwd] max |dlogp| vs the oracle in train mode: {err:.2e} (max|logp| {want_out.abs().max().item():.1f})")
    assert err < 2e-4 * scale
    loss = exit_ctc_losses(out, tgt, tl).sum()
    assert abs(loss.item() - want_loss.item()) < 2e-4 * max(1.0, abs(want_loss.item()))
    loss.backward()
    compare_grads(grads_of(gpu), grads_of(ref), 3e-3, f"{which} bf16x3")
    for (n, b_ref), (_, b_gpu) in zip(ref.named_buffers(), gpu.named_buffers()):
        if "running_" in n or "num_batches" in n:
            assert torch.allclose(b_gpu.cpu().float(), b_ref.float(), rtol=1e-4, atol=1e-6), n


def test_frozen_encoder_keeps_train_mode_semantics():
    """model.train() with only linears.* trainable: the reference still normalises with BATCH statistics (and updates the
    running ones); so does the default here (the eval-semantics shortcut is an explicit opt-in, model.frozen_encoder_eval)."""
    kw = base_kwargs(**SMALL)
    ref, gpu = make_train_pair(kw, seed=43)
    for m in (ref, gpu):
        for n, p in m.named_parameters():
            p.requires_grad_(n.startswith("linears."))
    mel, lens = synth.synth_mel(3, 80, 131, seed=43), torch.tensor([131, 90, 57])
    tgt, tl = synth.synth_targets(3, 9, kw["dec_voc_size"], seed=43)
    want_out = ref(mel, lens)
    R.summed_exit_ctc_loss(want_out, tgt, tl).backward()
    out = gpu(mel.cuda(), lens)
    assert (out.detach().cpu() - want_out.detach()).abs().max().item() < 2e-4
    exit_ctc_losses(out, tgt, tl).sum().backward()
    for e in range(kw["n_enc_exits"]):
        gw = ref.linears[e].weight.grad
        assert (gpu.linears[e].weight.grad.cpu() - gw).abs().max().item() < 2e-3 * gw.abs().max().item()
    assert all(p.grad is None for n, p in gpu.named_parameters() if not n.startswith("linears."))
    for (n, b_ref), (_, b_gpu) in zip(ref.named_buffers(), gpu.named_buffers()):
        if "running_" in n:
            assert torch.allclose(b_gpu.cpu().float(), b_ref.float(), rtol=1e-4, atol=1e-6), n


def test_data_parallel_buckets_hold_the_same_gradients():
    """enable_data_parallel (one rank: no collective): the backward writes every gradient into the flat per-exit-group buckets,
    p.grad ARE views of them, values equal the plain backward's bit for bit; a second backward onto existing gradients
    (no zero_grad) falls back to fresh tensors and accumulates like autograd always does."""
    kw = base_kwargs(**SMALL)
    _, gpu = make_train_pair(kw, seed=47)
    mel, lens = synth.synth_mel(3, 80, 131, seed=47).cuda(), torch.tensor([131, 90, 57])
    tgt, tl = synth.synth_targets(3, 9, kw["dec_voc_size"], seed=47)
    exit_ctc_losses(gpu(mel, lens), tgt, tl).sum().backward()
    plain = {n: p.grad.clone() for n, p in gpu.named_parameters()}
    gpu.zero_grad(set_to_none=True)
    for m in gpu.modules():  # the same batch statistics again: reset what the first step updated
        if isinstance(m, torch.nn.BatchNorm1d):
            m.reset_running_stats()
    gpu.enable_data_parallel(3, min_bucket_bytes=0)  # one bucket per exit group even on this small model
    exit_ctc_losses(gpu(mel, lens), tgt, tl).sum().backward()
    assert gpu.sync_gradients() == 0
    gb = gpu._dp["buckets"]
    assert [b["ready_after"] for b in gb.buckets][-1] == -1 and len(gb.buckets) >= 2
    for n, p in gpu.named_parameters():
        assert p.grad.data_ptr() == gb.view(n, p).data_ptr(), n
        assert torch.equal(p.grad, plain[n]), n
    exit_ctc_losses(gpu(mel, lens), tgt, tl).sum().backward()  # accumulates (autograd semantics), views untouched by the kernels
    for n, p in gpu.named_parameters():
        assert torch.allclose(p.grad, 2 * plain[n], rtol=1e-5, atol=1e-7), n


def _dp_train_rank(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share the one GPU of this box: gloo, not RCCL
    try:
        kw = base_kwargs(**SMALL)
        _, gpu = make_train_pair(kw, seed=59)
        B, T = 5, 131  # uneven shards: 3 + 2
        mel, lens = synth.synth_mel(B, 80, T, seed=59), torch.tensor([131, 90, 57, 131, 100])
        tgt, tl = synth.synth_targets(B, 9, kw["dec_voc_size"], seed=59)
        lo, hi = (0, 3) if rank == 0 else (3, 5)

        def local_step():
            gpu.zero_grad(set_to_none=True)
            for m in gpu.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.reset_running_stats()
            exit_ctc_losses(gpu(mel[lo:hi].cuda(), lens[lo:hi]), tgt[lo:hi], tl[lo:hi]).sum().backward()

        local_step()  # plain local gradients first
        mine = {n: p.grad.clone() for n, p in gpu.named_parameters()}
        gpu.enable_data_parallel(hi - lo, min_bucket_bytes=0)
        local_step()  # ... then the same step with the buckets: reduced group by group from the backward's callback
        n_coll = gpu.sync_gradients()
        torch.cuda.synchronize()
        # what the reduction must give: sum_r (B_r / B) * grad_r, checked on every parameter through an exchange of the local ones
        worst = 0.0
        for n, p in gpu.named_parameters():
            parts = [torch.empty_like(mine[n]).cpu() for _ in range(world)]
            dist.all_gather(parts, mine[n].cpu())
            want = sum(parts[r] * ((3, 2)[r] / 5.0) for r in range(world))
            worst = max(worst, ((p.grad.cpu() - want).abs().max() / (want.abs().max() + 1e-12)).item())
        in_views = all(p.grad.data_ptr() == gpu._dp["buckets"].view(n, p).data_ptr() for n, p in gpu.named_parameters())
        if rank == 0:
            q.put((n_coll, worst, in_views))
    finally:
        dist.destroy_process_group()


def test_world2_data_parallel_training_step_on_the_product_path():
    """BASELINE configs[3]'s mechanism with two ranks on this box's one GPU (gloo): the product module in train mode, uneven
    utterance shards, enable_data_parallel -> each exit group's flat bucket is all-reduced from eec_train_backward_ex's callback
    while the backward goes on -> sync_gradients.  Every parameter's gradient equals the shard-weighted sum of the ranks' local
    gradients (BatchNorm statistics stay per replica, as documented), and still lives in its bucket view."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_train_rank, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    n_coll, worst, in_views = q.get(timeout=300)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert n_coll == 3 and in_views  # exit group 1, exit group 0, stem
    assert worst < 1e-5, worst


def _rccl_single_rank(port, q):
    import os
    import torch.distributed as dist
    from early_exit_transformer_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        parallel.SINGLE_RANK_COLLECTIVES = True
        kw = base_kwargs(**SMALL)
        _, gpu = make_train_pair(kw, seed=61)
        B, T = 4, 131
        mel, lens = synth.synth_mel(B, 80, T, seed=61).cuda(), torch.tensor([131, 90, 57, 131])
        tgt, tl = synth.synth_targets(B, 9, kw["dec_voc_size"], seed=61)

        def step():
            gpu.zero_grad(set_to_none=True)
            for m in gpu.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.reset_running_stats()
            losses = exit_ctc_losses(gpu(mel, lens), tgt, tl)
            losses.sum().backward()
            return parallel.combine_exit_losses(losses.detach(), B)

        parallel.SINGLE_RANK_COLLECTIVES = False
        l0 = step()
        plain = {n: p.grad.clone() for n, p in gpu.named_parameters()}
        parallel.SINGLE_RANK_COLLECTIVES = True
        gpu.enable_data_parallel(B, min_bucket_bytes=0)
        l1 = step()  # the buckets' all-reduces leave from the backward's callback, through RCCL, with one rank
        n_coll = gpu.sync_gradients()
        torch.cuda.synchronize()
        same = all(torch.equal(p.grad, plain[n]) for n, p in gpu.named_parameters())
        q.put((n_coll, same, torch.allclose(l0, l1, rtol=1e-6, atol=0), dist.get_backend()))
    finally:
        dist.destroy_process_group()


def test_single_rank_rccl_runs_every_collective_of_the_training_step():
    """The real backend on a one-GPU box: a process group of ONE rank on "nccl" (= RCCL), parallel.SINGLE_RANK_COLLECTIVES, and
    the data-parallel training step of the product module.  The shard-size exchange, the loss all-reduce and the three bucket
    all-reduces (started from eec_train_backward_ex's callback, joined by sync_gradients) all go through RCCL's work objects
    and stream hand-over; with one rank the results must be those of the plain step, bit for bit."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank, args=(37500 + os.getpid() % 2000, q))
    p.start()
    n_coll, same, loss_ok, backend = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0
    assert backend == "nccl" and n_coll == 3 and same and loss_ok


@pytest.mark.parametrize("cfg", [SMALL, dict(SMALL, n_head=2), dict(SMALL, d_model=256, n_head=8)],
                         ids=["head_dim_16_unfused_attention", "head_dim_32_fused_attention", "d_model_256_fused_feed_forward"])
def test_dropout_masks_are_consistent_between_forward_and_backward(cfg):
    """drop_prob > 0: streams cannot match torch's, so the check is internal -- the same seed reproduces the step, another
    seed changes it, and the analytic gradient matches a central finite difference of the SAME masked network along a
    random direction in parameter space.  Both attention paths: batched GEMMs + softmax kernels (head dim 16) and the fused
    kernels that regenerate the attention-probability mask in three places (head dim 32).  d_model 256: the feed-forward modules
    run as one launch per direction (csrc/ffn.hip TR variants; d_ff 160 = one full chunk of 128 hidden units and a 32-wide rest) --
    the forward applies both of the module's masks, the backward regenerates the activation's."""
    kw = base_kwargs(**cfg)
    _, gpu = make_train_pair(kw, seed=5, drop=0.1)
    mel, lens = synth.synth_mel(2, 80, 99, seed=5).cuda(), torch.tensor([99, 70])
    tgt, tl = synth.synth_targets(2, 6, 32, seed=5)

    def step(seed):
        torch.manual_seed(seed)
        gpu.zero_grad()
        out = gpu(mel, lens)
        loss = exit_ctc_losses(out, tgt, tl).sum()
        loss.backward()
        return out.detach().clone(), loss.item(), {n: p.grad.clone() for n, p in gpu.named_parameters()}

    o1, l1, g1 = step(1)
    o2, l2, g2 = step(1)
    o3, l3, _ = step(2)
    assert torch.equal(o1, o2) and l1 == l2 and all(torch.equal(g1[n], g2[n]) for n in g1)
    assert not torch.equal(o1, o3)
    gpu.dropout = 0.0
    o0 = gpu(mel, lens).detach()
    gpu.dropout = 0.1
    assert torch.isfinite(o1).all() and (o1 - o0).abs().max().item() > 1e-3  # dropout is active

    gen = torch.Generator().manual_seed(0)
    names = ["conformer.0.conformer_layers.0.ffn1.sequential.1.weight", "conformer.1.conformer_layers.0.ffn2.sequential.4.weight",
             "conformer.1.conformer_layers.1.ffn1.sequential.0.weight",  # W1 and W2 of feed-forward modules, and a module's LayerNorm
             "conformer.1.conformer_layers.1.self_attn.in_proj_weight",
             "conformer.0.conformer_layers.1.conv_module.sequential.2.weight", "conv_subsample.sequential.1.weight", "linears.1.bias"]
    params = dict(gpu.named_parameters())
    for n in names:
        p = params[n]
        d = torch.randn(p.shape, generator=gen).cuda()
        d = d / d.norm()
        eps = 2e-2
        with torch.no_grad():
            p.add_(eps * d)
        lp = step(1)[1]
        with torch.no_grad():
            p.sub_(2 * eps * d)
        lm = step(1)[1]
        with torch.no_grad():
            p.add_(eps * d)
        fd = (lp - lm) / (2 * eps)
        an = (g1[n] * d).sum().item()
        print(f"\n[dropout grad check] {n}: analytic {an:.5e}  finite difference {fd:.5e}")
        assert abs(fd - an) < 0.05 * max(abs(an), abs(fd)) + 2e-3, n


def test_fused_feed_forward_equals_the_gemm_path_at_the_benchmark_shape():
    """The feed-forward modules of the training step run as one launch per direction (csrc/ffn.hip TR variants); EEC_TRAIN_FFN_FUSED=0 /
    EEC_TRAIN_FFN_FUSED_BWD=0 keep the LayerNorm + GEMM path of rounds 2-3.  Same seed -> same dropout masks on both paths, so the
    two steps are comparable WITH dropout at a size the CPU oracle cannot reach: the default 12-layer model, d_ff 2048, T = 1027,
    ragged lengths, 16 utterances.  Forward on fp16 pairs vs bf16 pairs and a different summation order: log-probs within 2e-4,
    every gradient within 2e-3 of its largest entry."""
    import os
    kw = base_kwargs()
    _, gpu = make_train_pair(kw, seed=77, drop=0.1)
    B, T = 16, 1027
    mel = synth.synth_mel(B, 80, T, seed=77).cuda()
    lens = torch.tensor([1027, 1027, 1000, 903, 771, 642, 515, 400, 1027, 259, 131, 99, 64, 47, 31, 31])
    tgt, tl = synth.synth_targets(B, 3, kw["dec_voc_size"], seed=77)  # 3 tokens: feasible for the 6-frame utterances too

    def step():
        torch.manual_seed(3)
        gpu.zero_grad()
        out = gpu(mel, lens)
        loss = exit_ctc_losses(out, tgt, tl).sum()
        loss.backward()
        return out.detach().cpu(), loss.item(), {n: p.grad.detach().cpu().double() for n, p in gpu.named_parameters()}

    saved = {k: os.environ.get(k) for k in ("EEC_TRAIN_FFN_FUSED", "EEC_TRAIN_FFN_FUSED_BWD")}
    try:
        os.environ.pop("EEC_TRAIN_FFN_FUSED", None), os.environ.pop("EEC_TRAIN_FFN_FUSED_BWD", None)
        o_f, l_f, g_f = step()
        os.environ["EEC_TRAIN_FFN_FUSED"] = "0"
        os.environ["EEC_TRAIN_FFN_FUSED_BWD"] = "0"
        o_g, l_g, g_g = step()
        os.environ["EEC_TRAIN_FFN_FUSED"] = "1"  # fused forward, GEMM-path backward: the tape the fused forward writes is the GEMM path's
        o_m, l_m, g_m = step()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    assert torch.isfinite(o_f).all() and not torch.equal(o_f, o_g)  # two different code paths did run
    err = (o_f - o_g).abs().max().item()
    print(f"\n[fused vs GEMM path, dropout 0.1] max |dlogp| {err:.2e}; losses {l_f:.6f} / {l_g:.6f}")
    assert err < 2e-4 and abs(l_f - l_g) < 2e-4 * abs(l_g)
    assert torch.equal(o_m, o_f)
    compare_grads(g_f, g_g, 2e-3, "fused feed-forward vs GEMM path")
    compare_grads(g_m, g_g, 2e-3, "fused forward + GEMM-path backward vs GEMM path")


def test_train_mode_without_autograd_keeps_train_semantics():
    """model.train() under torch.no_grad() (the reference would still use batch statistics and dropout): same output as the
    autograd forward with the same seed, different from eval mode."""
    kw = base_kwargs(n_enc_exits=1, n_enc_layers=1, d_feed_forward=256)
    _, gpu = make_train_pair(kw, seed=3, drop=0.1)
    mel, lens = synth.synth_mel(2, 80, 99, seed=3).cuda(), torch.tensor([99, 70])
    torch.manual_seed(4)
    with_grad = gpu(mel, lens).detach()
    torch.manual_seed(4)
    with torch.no_grad():
        without = gpu(mel, lens)
    assert torch.equal(with_grad, without) and not without.requires_grad
    gpu.eval()
    with torch.no_grad():
        ev = gpu(mel, lens)
    assert (ev - without).abs().max().item() > 1e-3


ref_decode = ref_decoder_logits  # the reference's decoder arithmetic through the torch modules (tests/conftest.py), CPU side only


def make_aed_pair(kw, common, seed):
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    import aed_fixture as G
    gpu = full_conformer(device="cuda", **common, **kw)
    sd = G.aed_state_dict(gpu, seed)
    gpu.load_state_dict(sd, strict=True)
    cpu = full_conformer(device="cpu", **common, **kw)
    cpu.load_state_dict(sd, strict=True)
    return cpu.train(), gpu.cuda().train(), sd


@pytest.mark.parametrize("B,S,Tq,n_dec,d_model,n_head", [(3, 7, 31, 2, 256, 8), (4, 42, 64, 3, 256, 8), (2, 19, 40, 2, 128, 2), (5, 1, 9, 1, 64, 4)])
def test_decoder_training_step_matches_reference_modules(B, S, Tq, n_dec, d_model, n_head):
    """The AED decoder alone (eec_decoder_train_forward / _backward behind model._decode_one in train mode) against torch autograd
    through the reference's modules on the CPU, drop_prob 0: logits, the gradient of every decoder parameter (embedding table and
    the shared final LayerNorm included) and the gradient of the encoder output it attended to; padded target positions, head
    dims 32 / 64 / 16, a single-token prefix."""
    kw = dict(n_enc_exits=2, n_enc_layers=1, d_model=d_model, n_head=n_head, d_feed_forward=192, depthwise_kernel_size=7, dec_voc_size=64)
    common = dict(trg_pad_idx=30, enc_voc_size=64, max_len=400, features_length=80, drop_prob=0.0, n_dec_layers=n_dec)
    cpu, gpu, _ = make_aed_pair(kw, common, seed=13)
    g = torch.Generator().manual_seed(B * 100 + S)
    trg = torch.randint(3, 64, (B, S), generator=g)
    trg[:, 0] = 1
    if S > 4:
        trg[1, S - 3:] = 30  # padding at the end of one target
    enc = torch.randn(B, Tq, d_model, generator=g)
    w = torch.randn(B, S, 64, generator=g)  # a fixed linear functional of the logits as the loss
    for idx in (1, 0):
        e_ref = enc.clone().requires_grad_(True)
        want = ref_decode(cpu, trg, e_ref, idx)
        cpu.zero_grad()
        (want * w).sum().backward()
        e_gpu = enc.cuda().requires_grad_(True)
        got = gpu._decode_one(trg.cuda(), e_gpu, idx)
        assert got.requires_grad and got.shape == want.shape
        err = (got.detach().cpu() - want.detach()).abs().max().item()
        assert err < 2e-4 * max(1.0, want.detach().abs().max().item()), err
        gpu.zero_grad()
        (got * w.cuda()).sum().backward()
        wantg = {n: p.grad.double() for n, p in cpu.named_parameters() if p.grad is not None}
        gotg = {n: p.grad.detach().cpu().double() for n, p in gpu.named_parameters() if p.grad is not None}
        assert set(wantg) == set(gotg), sorted(set(wantg) ^ set(gotg))[:5]
        compare_grads(gotg, wantg, 2e-3, f"decoder {idx} bf16x3")
        ge, gw = e_gpu.grad.cpu(), e_ref.grad
        assert (ge - gw).abs().max().item() < 2e-3 * gw.abs().max().item(), "gradient of the encoder output"


def test_decoder_dropout_masks_are_consistent_between_forward_and_backward():
    """drop_prob > 0: torch's streams cannot match, so the check is internal -- one seed reproduces the logits, another changes
    them, eval-mode inference is untouched, and the analytic gradient equals a central finite difference of the SAME masked
    network along a random direction in (parameter, encoder-output) space."""
    kw = dict(n_enc_exits=2, n_enc_layers=1, d_model=128, n_head=4, d_feed_forward=160, depthwise_kernel_size=7, dec_voc_size=64)
    common = dict(trg_pad_idx=30, enc_voc_size=64, max_len=400, features_length=80, drop_prob=0.1, n_dec_layers=2)
    _, gpu, _ = make_aed_pair(kw, common, seed=17)
    g = torch.Generator().manual_seed(5)
    trg = torch.randint(3, 64, (3, 9), generator=g).cuda()
    enc = torch.randn(3, 21, 128, generator=g).cuda()
    w = torch.randn(3, 9, 64, generator=g).cuda()
    named = gpu._decoder_named_params(1)

    def f(seed, e=enc):
        return (gpu._decode_one(trg, e, 1, seed=seed) * w).mean()

    a, b, c = f(7).item(), f(7).item(), f(8).item()
    assert a == b and abs(a - c) > 1e-6 * abs(a)  # one seed reproduces the masked network, another draws other masks
    gpu.dropout = 0.0
    a0 = f(7).item()
    gpu.dropout = 0.1
    assert abs(a - a0) > 1e-6 * abs(a0)  # dropout is active
    e = enc.clone().requires_grad_(True)
    gpu.zero_grad()
    f(7, e).backward()
    params = dict(named)
    gen = torch.Generator().manual_seed(0)
    names = ["emb.weight", "decoders.1.layers.0.self_attn.in_proj_weight", "decoders.1.layers.1.multihead_attn.in_proj_weight",
             "decoders.1.layers.0.linear1.weight", "decoders.1.layers.1.linear2.weight", "decoders.1.layers.1.norm2.weight",
             "layer_norm.bias", "linears_2.1.weight", "<enc>"]
    for n in names:
        p = enc if n == "<enc>" else params[n]
        grad = e.grad if n == "<enc>" else p.grad
        d = torch.randn(p.shape, generator=gen).cuda()
        d = d / d.norm()
        eps = 2e-2
        if n == "<enc>":
            lp, lm = f(7, enc + eps * d).item(), f(7, enc - eps * d).item()
        else:
            with torch.no_grad():
                p.add_(eps * d)
            lp = f(7).item()
            with torch.no_grad():
                p.sub_(2 * eps * d)
            lm = f(7).item()
            with torch.no_grad():
                p.add_(eps * d)
        fd = (lp - lm) / (2 * eps)
        an = (grad * d).sum().item()
        print(f"\n[decoder dropout grad check] {n}: analytic {an:.5e}  finite difference {fd:.5e}")
        assert abs(fd - an) < 0.05 * max(abs(an), abs(fd)) + 2e-3, n


def test_aed_training_step_matches_reference_modules():
    """train.py:36-52 (decoder_mode aed) on full_conformer, everything on the HIP training kernels: the encoder's forward /
    backward (the taps it hands to the decoders are differentiable outputs of the same autograd function) and the decoders'
    (csrc/decoder_train.hip; no torch module runs under autograd -- checked by making their forward raise).
    Against the same parameters run through the oracle encoder + the reference's decoder modules on the CPU (drop_prob 0):
    CTC log-probs, decoder logits and the gradient of every parameter."""
    import os
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    import aed_fixture as G
    kw = dict(n_enc_exits=2, n_enc_layers=1, d_model=256, n_head=8, d_feed_forward=256, depthwise_kernel_size=15, dec_voc_size=64)
    common = dict(trg_pad_idx=30, enc_voc_size=64, max_len=400, features_length=80, drop_prob=0.0, n_dec_layers=2)
    gpu = full_conformer(device="cuda", **common, **kw)
    sd = G.aed_state_dict(gpu, 11)
    gpu.load_state_dict(sd, strict=True)
    cpu = full_conformer(device="cpu", **common, **kw)
    cpu.load_state_dict(sd, strict=True)
    ref_enc = R.EarlyConformerRef(**base_kwargs(drop_prob=0.0, dec_voc_size=64, enc_voc_size=64, max_len=400, **{k: v for k, v in kw.items() if k != "dec_voc_size"}))
    enc_sd = {k.replace("linears_1.", "linears.").replace("positional_encoder_1.", "positional_encoder."): v for k, v in sd.items()
              if k.startswith(("conv_subsample.", "conformer.", "linears_1.", "positional_encoder_1."))}
    ref_enc.load_state_dict(enc_sd, strict=True)
    ref_enc.train(), cpu.train(), gpu.cuda().train()
    B, T = 3, 131
    mel, lens = synth.synth_mel(B, 80, T, seed=2), torch.tensor([131, 100, 77])
    g = torch.Generator().manual_seed(3)
    trg = torch.randint(3, 64, (B, 7), generator=g)
    trg[:, 0] = 1
    trg[1, 5:] = 30
    tgt, tl = synth.synth_targets(B, 6, 64, seed=2)
    ce = torch.nn.CrossEntropyLoss(ignore_index=30)

    # reference side: oracle encoder (taps through a hook on the Conformer groups) + the decoder modules of `cpu`
    taps = []
    hooks = [m.register_forward_hook(lambda _m, _i, o: taps.append(o[0] if isinstance(o, tuple) else o)) for m in ref_enc.conformer]
    want_enc = ref_enc(mel, lens)
    [h.remove() for h in hooks]
    want_dec = torch.stack([ref_decode(cpu, trg[:, :-1], taps[e], e) for e in range(2)])
    want_loss = R.summed_exit_ctc_loss(want_enc, tgt, tl) + sum(ce(want_dec[e].reshape(-1, 64), trg[:, 1:].reshape(-1)) for e in range(2))
    want_loss.backward()

    def no_torch_decoder(*_a, **_k):
        raise AssertionError("a torch decoder module ran under autograd: the training step must stay on the HIP path")
    for m in list(gpu.decoders) + [gpu.emb, gpu.layer_norm] + list(gpu.linears_2):
        m.forward = no_torch_decoder
    dec_out, enc_out = gpu(mel.cuda(), lens, trg[:, :-1].cuda())
    assert (enc_out.detach().cpu() - want_enc.detach()).abs().max().item() < 2e-4
    assert (dec_out.detach().cpu() - want_dec.detach()).abs().max().item() < 2e-3
    loss = exit_ctc_losses(enc_out, tgt, tl).sum() + sum(ce(dec_out[e].reshape(-1, 64), trg[:, 1:].cuda().reshape(-1)) for e in range(2))
    assert abs(loss.item() - want_loss.item()) < 1e-3 * max(1.0, abs(want_loss.item()))
    loss.backward()
    want = {}
    for n, p in ref_enc.named_parameters():
        want[n.replace("linears.", "linears_1.")] = p.grad.double()
    for n, p in cpu.named_parameters():
        if p.grad is not None:
            want[n] = p.grad.double()
    got = {n: p.grad.detach().cpu().double() for n, p in gpu.named_parameters() if p.grad is not None}
    assert set(want) <= set(got), sorted(set(want) - set(got))[:5]
    compare_grads(got, want, 5e-3, "aed bf16x3")


def test_side_stream_weight_gradients_are_bit_identical_to_the_single_stream_order():
    """The weight-gradient GEMMs run on a second stream beside the dX chain.  The arithmetic is the same either way, so every
    gradient must be BIT-identical to a run with the jobs on the main stream (EEC_TRAIN_NO_SIDE=1), repeatedly, at a size where
    the two streams really overlap: a missed dependency shows up as a mismatch."""
    import os
    kw = base_kwargs(n_enc_exits=2, n_enc_layers=2, d_feed_forward=1024)
    _, gpu = make_train_pair(kw, seed=17, drop=0.1)
    mel, lens = synth.synth_mel(16, 80, 1027, seed=17).cuda(), torch.tensor([1027] * 8 + [900, 800, 700, 600, 500, 400, 300, 200])
    tgt, tl = synth.synth_targets(16, 20, 256, seed=17)

    def grads():
        torch.manual_seed(5)
        gpu.zero_grad(set_to_none=True)
        exit_ctc_losses(gpu(mel, lens), tgt, tl).sum().backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in gpu.named_parameters()}

    os.environ["EEC_TRAIN_NO_SIDE"] = "1"
    try:
        want = grads()
    finally:
        del os.environ["EEC_TRAIN_NO_SIDE"]
    for rep in range(6):
        got = grads()
        bad = [n for n in want if not torch.equal(got[n], want[n])]
        assert not bad, (rep, bad[:4])


def test_reference_training_loop_reduces_the_loss():
    """The loop of train.py:27-75 (AdamW, clip_grad_norm_, summed per-exit CTC loss) on the product module."""
    kw = base_kwargs(**SMALL)
    _, gpu = make_train_pair(kw, seed=9, drop=0.1)
    from oracle.conformer_ref import xavier_like_reference
    xavier_like_reference(gpu)
    mel, lens = synth.synth_mel(4, 80, 131, seed=9).cuda(), torch.tensor([131, 120, 100, 80])
    tgt, tl = synth.synth_targets(4, 8, 32, seed=9)
    opt = torch.optim.AdamW(gpu.parameters(), lr=2e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.1)
    torch.manual_seed(0)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        enc = gpu(mel, lens)
        loss = exit_ctc_losses(enc, tgt, tl).sum()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(gpu.parameters(), 1.0)
        opt.step()
        losses.append(loss.item())
    print("\n[train loop] summed exit CTC loss per step:", " ".join(f"{v:.3f}" for v in losses))
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < 0.8 * losses[0]
    # the eval path picks up the updated weights and running statistics
    gpu_eval = Early_conformer(**{**dict(base_kwargs(), d_feed_forward=512, n_enc_exits=1, n_enc_layers=1), "device": "cuda"}).cuda()
    gpu_eval.train()
    out = gpu_eval(synth.synth_mel(1, 80, 99, seed=1).cuda(), torch.tensor([99]))
    exit_ctc_losses(out, *synth.synth_targets(1, 5, 256, seed=1)).sum().backward()
    gpu_eval.eval()
    with torch.no_grad():
        assert torch.isfinite(gpu_eval(synth.synth_mel(1, 80, 99, seed=1).cuda(), torch.tensor([99]))).all()
