"""CPU tests of the host side: the C-ABI library loads and exports everything include/eec.h declares
(no compute calls without a GPU), argument validation, loud failure without a HIP device, and the
world_size-2 sharded-loss path over gloo."""
import ctypes as C
import os
import re

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, base_kwargs
from early_exit_transformer_amd import capi, parallel, synth
from early_exit_transformer_amd.build import LIB_PATH


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB_PATH):
        from early_exit_transformer_amd.build import build_library
        build_library()
    return capi.load()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "eec.h")).read()
    declared = set(re.findall(r"\b(eec_[a-z_0-9]+)\s*\(", header))
    declared -= {"eec_encoder"}  # the opaque struct tag
    assert declared, "no prototypes parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/eec.h but not exported by libeec.so"
    assert set(capi.EXPORTS) == declared
    assert lib.eec_abi_version() == 16


def test_loading_the_library_first_keeps_one_hip_runtime_in_the_process(lib):
    """build() then smoke() in one interpreter: dlopen of libeec.so before torch must not bring /opt/rocm's libamdhip64 in
    beside torch's bundled one (two runtimes in a process: the second reports no device)."""
    import subprocess
    import sys
    code = ("from early_exit_transformer_amd import capi; capi.load(); import torch\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
            "print(len(libs), libs)")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == "1", out.stdout


def test_trainer_workspace_sizing_runs_without_a_device(lib):
    """eec_trainer_workspace_bytes is host arithmetic (a dry run of the forward + backward carve): monotonic in the batch and
    of the size the saved activations predict (default model, B = 64, T = 1027: ~10 GB of fp32 tape + scratch)."""
    import ctypes as C
    cfg = capi.EecConfig(256, 8, 2048, 31, 6, 2, 80, 256, 2000, capi.ARCH_CONFORMER)
    h = C.c_void_p()
    assert lib.eec_trainer_create(C.byref(cfg), C.byref(h)) == 0, lib.eec_trainer_last_error()
    try:
        sizes = [lib.eec_trainer_workspace_bytes(h, b, 1027) for b in (1, 8, 64)]
        assert sizes[0] > 0 and sizes[0] < sizes[1] < sizes[2]
        M, D, F = 64 * 256, 256, 2048
        per_layer = 4 * (13 * M * D + 4 * M * F + 3 * M * D + 2 * M * D)  # the saved activations of one layer (fused attention: no P)
        assert 0.9 * 12 * per_layer < sizes[2] < 1.25 * 12 * per_layer
        assert lib.eec_trainer_workspace_bytes(h, 4, 5) == 0  # T too short
    finally:
        lib.eec_trainer_destroy(h)


def test_decoder_workspace_sizing_and_argument_checks(lib):
    """eec_decoder_workspace_bytes is host arithmetic; eec_decoder_forward rejects bad arguments before touching a device."""
    import ctypes as C
    n1 = lib.eec_decoder_workspace_bytes(256, 8, 2048, 256, 10, 5, 256)
    n2 = lib.eec_decoder_workspace_bytes(256, 8, 2048, 256, 10, 50, 256)
    assert 0 < n1 < n2
    assert lib.eec_decoder_workspace_bytes(256, 7, 2048, 256, 10, 5, 256) == 0  # d_model not divisible by the heads
    assert lib.eec_decoder_workspace_bytes(256, 8, 2048, 256, 0, 5, 256) == 0
    ps = capi.EecDecoderParams()
    rc = lib.eec_decoder_forward(C.byref(ps), 256, 8, 2048, 256, 126, None, None, 1, 1, 1, 0, 3, 1, None, None, 0, None)
    assert rc != 0 and b"null" in lib.eec_decoder_last_error()


def test_decoder_step_cache_sizing_and_argument_checks(lib):
    """The step-wise decoder's cache size is host arithmetic (linear in the steps and in T'); unsupported geometries size to
    0 and the entry points reject bad arguments before touching a device."""
    import ctypes as C
    assert lib.eec_decoder_step_max_beams() == 16
    n1 = lib.eec_decoder_cache_bytes(256, 8, 2048, 256, 6, 40, 256)
    n2 = lib.eec_decoder_cache_bytes(256, 8, 2048, 256, 6, 80, 256)
    n3 = lib.eec_decoder_cache_bytes(256, 8, 2048, 256, 6, 80, 512)
    assert 0 < n1 < n2 < n3
    per_step = 6 * 16 * 2 * 256 * 4  # keys | values of 16 beam slots in every layer
    assert abs((n2 - n1) - 40 * (per_step + 2 * 16 * 4 + 16)) <= 4096
    assert lib.eec_decoder_cache_bytes(512, 8, 2048, 256, 6, 40, 256) > n1   # head dim 64
    assert lib.eec_decoder_cache_bytes(384, 8, 2048, 256, 6, 40, 256) == 0   # head dim 48: not a power-of-two share of a wave
    assert lib.eec_decoder_cache_bytes(256, 8, 4096, 256, 6, 40, 256) == 0   # 16 rows of d_ff do not fit the LDS
    ps = capi.EecDecoderParams()
    assert lib.eec_decoder_begin(C.byref(ps), 256, 8, 2048, 256, None, 256, 40, 3, None, 0, None) != 0
    assert b"null" in lib.eec_decoder_step_last_error()
    rc = lib.eec_decoder_step(C.byref(ps), 256, 8, 2048, 256, 126, None, None, 1, 0, 0, 256, 40, 1, None, None, 0, None)
    assert rc != 0 and b"null" in lib.eec_decoder_step_last_error()


def test_out_frames_matches_conv_arithmetic(lib):
    for T in (7, 8, 10, 11, 131, 1027, 2051, 8003):
        t1 = (T - 3) // 2 + 1
        assert lib.eec_out_frames(T) == (t1 - 3) // 2 + 1
    assert lib.eec_out_frames(6) == 0
    assert lib.eec_out_frames(1027) == 256


@pytest.mark.parametrize("field,value,code", [("d_model", 128, 10002), ("n_heads", 3, 10001), ("d_ff", 100, 10002), ("n_mels", 40, 10002),
                                              ("dw_kernel", 32, 10002), ("dw_kernel", 33, 10002), ("vocab", 300, 10002),
                                              ("n_exits", 0, 10001)])
def test_create_rejects_unsupported_configs_before_touching_the_gpu(lib, field, value, code):
    cfg = capi.EecConfig(256, 8, 2048, 31, 6, 2, 80, 256, 2000, 0)
    setattr(cfg, field, value)
    h = C.c_void_p()
    assert lib.eec_encoder_create(C.byref(cfg), C.byref(h)) == code
    assert lib.eec_last_error()


def test_product_module_has_no_cpu_path():
    from early_exit_transformer_amd.model import Early_conformer
    from early_exit_transformer_amd.conformer import Conformer
    m = Early_conformer(**base_kwargs(n_enc_exits=1, n_enc_layers=1, d_feed_forward=128)).eval()
    with pytest.raises(RuntimeError, match="HIP device only"):
        m(torch.zeros(1, 80, 64), torch.tensor([64]))
    with pytest.raises(RuntimeError, match="parameter container"):
        Conformer(256, 8, 128, 1, 31)(torch.zeros(1, 4, 256), torch.tensor([4]))
    from early_exit_transformer_amd.model import Early_zipformer, Splitformer
    for cls, n in ((Splitformer, 2), (Early_zipformer, 19)):
        other = cls(**base_kwargs(n_enc_exits=n, n_enc_layers=1, d_feed_forward=128)).eval()
        with pytest.raises(RuntimeError, match="HIP device only"):
            other(torch.zeros(1, 80, 64), torch.tensor([64]))
    with pytest.raises(ValueError):
        Early_zipformer(**base_kwargs(n_enc_exits=6, n_enc_layers=1, d_feed_forward=128))
    import early_exit_transformer_amd.model as pm
    src = open(pm.__file__).read()
    assert "oracle" not in src.replace("oracle/", "") or "import oracle" not in src


def test_shard_range_partitions():
    for n in (1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def _rank_main(rank, world, port, q):
    from oracle import conformer_ref as R
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kw = base_kwargs(n_enc_exits=2, n_enc_layers=1, d_feed_forward=128)
        m = R.EarlyConformerRef(**kw).eval()
        m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0, style="trained"))
        B, T = 5, 131  # uneven shards: 3 + 2
        mel = synth.synth_mel(B, 80, T, seed=0)
        lens = torch.tensor([131, 120, 111, 131, 80])  # every shard holds a full-length utterance (a3 invariant)
        tgt, tl = synth.synth_targets(B, 10, 256, seed=0)
        ctc = torch.nn.CTCLoss(blank=0, reduction="mean", zero_infinity=True)

        def exit_losses(lp, t, l):
            il = torch.full((lp.size(1),), lp.size(2), dtype=torch.long)
            return torch.stack([ctc(lp[e].permute(1, 0, 2), t, il, l) for e in range(lp.size(0))])

        lo, hi = parallel.shard_range(B, rank, world)
        with torch.no_grad():
            local = exit_losses(m(mel[lo:hi], lens[lo:hi]), tgt[lo:hi], tl[lo:hi])
            combined = parallel.combine_exit_losses(local, hi - lo)
            if rank == 0:
                full = exit_losses(m(mel, lens), tgt, tl)
                q.put((combined.tolist(), full.tolist()))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_sharded_exit_loss_equals_global_batch():
    """N>1 path: utterance shards + one all-reduce reproduce the single-process batch-mean CTC loss."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    combined, full = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert combined == pytest.approx(full, rel=2e-5, abs=2e-5)


def _grad_rank_main(rank, world, port, q):
    from oracle import conformer_ref as R
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kw = base_kwargs(n_enc_exits=2, n_enc_layers=1, d_feed_forward=128, d_model=64, n_head=4)
        m = R.EarlyConformerRef(**kw).eval()  # running BatchNorm statistics: shards are then independent (per-replica batch
        m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0, style="trained"))  # statistics are the documented choice)
        B, T = 5, 99
        mel, lens = synth.synth_mel(B, 80, T, seed=0), torch.tensor([99, 90, 80, 99, 70])
        tgt, tl = synth.synth_targets(B, 8, 256, seed=0)
        lo, hi = parallel.shard_range(B, rank, world)
        R.summed_exit_ctc_loss(m(mel[lo:hi], lens[lo:hi]), tgt[lo:hi], tl[lo:hi]).backward()
        n_coll = parallel.allreduce_gradients(list(m.parameters()), hi - lo, bucket_bytes=100 << 10)
        got = [p.grad.clone() for p in m.parameters()]
        # a LAST PARTIAL BATCH: 3 + 2 utterances above, 2 + 2 now -- rank 1's own size is unchanged, rank 0's is not.  Every rank
        # must still enter the same collectives (a weight cached per rank-local size would let rank 1 skip the exchange) and
        # the weights must follow the new sizes
        m.zero_grad()
        lo2, hi2 = (0, 2) if rank == 0 else (2, 4)
        R.summed_exit_ctc_loss(m(mel[lo2:hi2], lens[lo2:hi2]), tgt[lo2:hi2], tl[lo2:hi2]).backward()
        parallel.allreduce_gradients(list(m.parameters()), hi2 - lo2, bucket_bytes=100 << 10)
        got2 = [p.grad.clone() for p in m.parameters()]
        if rank == 0:
            rel = lambda got: max(((g - p.grad).abs().max() / (p.grad.abs().max() + 1e-6)).item() for g, p in zip(got, m.parameters()))  # noqa: E731
            m.zero_grad()
            R.summed_exit_ctc_loss(m(mel, lens), tgt, tl).backward()
            err = rel(got)
            m.zero_grad()
            R.summed_exit_ctc_loss(m(mel[:4], lens[:4]), tgt[:4], tl[:4]).backward()
            q.put((n_coll, max(err, rel(got2))))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_bucketed_gradient_allreduce_equals_global_batch_gradient():
    """Config-4 path: per-rank backward on an utterance shard + bucketed, batch-weighted all-reduce == the gradient of the
    global-batch mean loss."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_grad_rank_main, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    n_coll, err = q.get(timeout=180)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert n_coll >= 2  # more than one bucket (the shard-size exchange is a collective of its own, on every call)
    assert err < 1e-4


def _bucket_rank_main(rank, world, port, q):
    from oracle import conformer_ref as R
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kw = base_kwargs(n_enc_exits=2, n_enc_layers=1, d_feed_forward=128, d_model=64, n_head=4)
        m = R.EarlyConformerRef(**kw).eval()
        m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0, style="trained"))
        B, T = 6, 99
        mel, lens = synth.synth_mel(B, 80, T, seed=0), torch.tensor([99, 90, 80, 85, 99, 60])  # every shard holds a full-length utterance
        tgt, tl = synth.synth_targets(B, 8, 256, seed=0)
        lo, hi = (0, 4) if rank == 0 else (4, 6)  # unequal shards: the weights b_r / B matter
        named = list(m.named_parameters())
        gb = parallel.GradBuckets(named, n_groups=2, min_bucket_bytes=0)
        order = [b["ready_after"] for b in gb.buckets]
        w = parallel.shard_weight(hi - lo, torch.device("cpu"))
        assert w == pytest.approx((hi - lo) / 6.0)
        # never cached: a second call is a second exchange every rank enters (same answer)
        assert parallel.shard_weight(hi - lo, torch.device("cpu")) == w

        def local_backward():
            m.zero_grad(set_to_none=True)
            R.summed_exit_ctc_loss(m(mel[lo:hi], lens[lo:hi]), tgt[lo:hi], tl[lo:hi]).backward()

        # (1) the training backward's protocol: gradients written into the views, buckets reduced as their group finishes
        local_backward()
        for n, p in named:
            v = gb.view(n, p)
            v.copy_(p.grad)
            p.grad = v
        for e in (1, 0, -1):
            for i in gb.buckets_ready_after(e):
                gb.allreduce_bucket(i, w, trusted=True)
        n_wait = gb.wait()
        got_views = [p.grad.clone() for _, p in named]
        in_place = all(p.grad.data_ptr() == gb.view(n, p).data_ptr() for n, p in named)
        # (2) gradients that live outside the buckets (accumulated into older tensors): the gather / scatter path
        local_backward()
        n_all = gb.allreduce_all(w)
        got_plain = [p.grad.clone() for _, p in named]
        # (3) equal-shard loss combination: mean of the local means, no count exchanged
        eq = parallel.combine_exit_losses(torch.tensor([1.0 + rank, 3.0 - rank]), 5, equal_shards=True)
        if rank == 0:
            m.zero_grad(set_to_none=True)
            R.summed_exit_ctc_loss(m(mel, lens), tgt, tl).backward()
            ref = [p.grad for _, p in named]
            err = lambda got: max(((g - r).abs().max() / (r.abs().max() + 1e-6)).item() for g, r in zip(got, ref))  # noqa: E731
            q.put((order, n_wait, n_all, in_place, err(got_views), err(got_plain), eq.tolist()))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_flat_gradient_buckets_reduce_in_place_per_exit_group():
    """parallel.GradBuckets: one flat bucket per exit group in the order the backward finishes them (last group first, stem
    last); reducing the buckets as the groups finish, in place, equals the gradient of the global-batch mean loss with unequal
    shards; gradients outside the buckets take the gather path to the same result; the shard weight is exchanged once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bucket_rank_main, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    order, n_wait, n_all, in_place, err_views, err_plain, eq = q.get(timeout=180)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert order == [1, 0, -1] and n_wait == 3 and n_all == 3 and in_place
    assert err_views < 1e-4 and err_plain < 1e-4
    assert eq == pytest.approx([1.5, 2.5])


def _sync_rank_main(rank, world, port, q):
    """sync_gradients() of a model whose backward never reports into the buckets (Splitformer / Early_zipformer / the heads-only
    step go through autograd functions that know nothing of _dp): every bucket must still be reduced."""
    import types
    from early_exit_transformer_amd.model import _HipEncoderMixin
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        class Toy(_HipEncoderMixin, torch.nn.Module):  # the mixin's data-parallel half needs only _cfg.n_exits and parameters
            def __init__(self):
                torch.nn.Module.__init__(self)
                self.conformer = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(2)])
                self.linears = torch.nn.ModuleList([torch.nn.Linear(8, 4) for _ in range(2)])
                self.stem = torch.nn.Linear(3, 8)
                self._cfg = types.SimpleNamespace(n_exits=2)
                self._enc = self._trainer = None
        torch.manual_seed(0)
        m = Toy()
        b_local = 3 if rank == 0 else 1
        m.enable_data_parallel(b_local, min_bucket_bytes=0)
        # (1) plain autograd gradients, nothing reduced from a callback
        for i, p in enumerate(m.parameters()):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        n1 = m.sync_gradients()
        want = [(3 * 1 + 1 * 2) / 4.0 * (i + 1) for i in range(len(list(m.parameters())))]
        ok1 = all(torch.allclose(p.grad, torch.full_like(p, w)) for p, w in zip(m.parameters(), want))
        # (2) a bucket reduced from the backward's callback whose views autograd did NOT adopt (a hook made it clone): p.grad is
        # a stale copy taken before the reduction -> sync_gradients() repairs it from the reduced view
        dp = m._dp
        gb = dp["buckets"]
        named = dict(m.named_parameters())
        for p in m.parameters():
            p.grad = None
        i0 = 0
        for n, p in gb.buckets[i0]["params"]:
            gb.view(n, p).fill_(float(rank + 1))
        gb.allreduce_bucket(i0, dp["weight"], dp["group"], trusted=True)
        dp["reduced"] = {i0}
        for n, p in gb.buckets[i0]["params"]:
            p.grad = torch.full_like(p, float(rank + 1))  # the unreduced copy
        for i in range(1, len(gb.buckets)):
            for n, p in gb.buckets[i]["params"]:
                p.grad = torch.full_like(p, float(rank + 1))
        n2 = m.sync_gradients()
        ok2 = all(torch.allclose(p.grad, torch.full_like(p, (3 * 1 + 1 * 2) / 4.0)) for p in m.parameters())
        if rank == 0:
            q.put((n1, ok1, n2, ok2, len(gb.buckets), dp["reduced"] == set()))
        assert named
    finally:
        dist.destroy_process_group()


def test_world2_gloo_sync_gradients_reduces_buckets_the_backward_never_reported():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_sync_rank_main, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    n1, ok1, n2, ok2, nb, cleared = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert nb == 3 and n1 == nb and ok1
    assert n2 == nb and ok2 and cleared


def _run_bench(extra_env, *argv):
    import subprocess
    import sys
    env = dict(os.environ, EEC_BENCH_PLUMBING="1", **extra_env)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=300)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (no torchrun): the parent starts N rank processes itself, relays rank 0's ONE JSON line
    and the ranks really exchanged data (the combined value is the B-weighted mean of both ranks' inputs)."""
    import json
    res = _run_bench({}, "--gpus", "2", "--steps", "2", "--batch", "64")
    assert res.returncode == 0, res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["plumbing"] is True
    want = [(e * 64 + (e + 1) * 65) / 129.0 for e in range(6)]  # rank r holds exit values e + r with B_r = 64 + r
    assert rec["combined"] == pytest.approx(want, rel=1e-6)


def test_bench_launcher_propagates_a_failed_rank():
    res = _run_bench({"EEC_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "1")
    assert res.returncode == 7
    assert "rank 1 exited with status 7" in res.stderr


def test_bench_rank_failing_inside_a_step_fails_the_job():
    """A rank that raises inside a training step skips collectives its peers wait in.  It must leave with a non-zero status
    (never a record with rc 0): the launcher then terminates the stuck peers and propagates the status; when the failing rank
    is rank 0 the record is still emitted once, with the error in place of the training-step lines."""
    import json
    res = _run_bench({"EEC_BENCH_FAIL_STEP_RANK": "1", "EEC_BENCH_TRAIN_TIMEOUT": "60"}, "--gpus", "2", "--steps", "1")
    assert res.returncode == 4, (res.returncode, res.stderr)
    assert "rank 1 exited with status 4" in res.stderr
    # rank 0 was blocked in a collective when the launcher terminated it: its SIGTERM thread still emits the ONE record
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    assert "peer rank failed" in json.loads(lines[0])["train_step"]["error"]
    res = _run_bench({"EEC_BENCH_FAIL_STEP_RANK": "0", "EEC_BENCH_TRAIN_TIMEOUT": "60"}, "--gpus", "2", "--steps", "1")
    assert res.returncode == 4, (res.returncode, res.stderr)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "injected failure" in json.loads(lines[0])["train_step"]["error"]


def test_bench_hung_section_ends_non_zero():
    """The watchdog of the last section ends a hung run with a NON-ZERO status after emitting the record."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    guard = src[src.index("class RecordGuard"):src.index("def plumbing_check")]
    assert "os._exit(0)" not in src and "os._exit(code)" in guard and ", 3)" in guard


def test_bench_parent_makes_no_gpu_call_before_spawning():
    """The launcher branch must come before anything that initialises HIP (torch.cuda.*), or the ranks inherit a
    process that already owns the device."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args.gpus") < main.index("torch.cuda.")
    launcher = src[src.index("def launch_ranks"):src.index("def plumbing_check")]
    assert "torch.cuda" not in launcher and "import torch" not in launcher


class _ToyDecoderModel:
    """A deterministic stand-in for ``model._decoder_``: log-probs depend on the whole prefix, so beam bookkeeping errors
    (wrong parent beam, wrong order) change the result."""

    def __init__(self, V=11, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.emb = torch.randn(V, 16, generator=g)
        self.out = torch.randn(16, V, generator=g) * 2

    def _head(self, h, layer_n):
        return torch.log_softmax(torch.tanh(h) @ self.out * (1.0 + 0.1 * layer_n), dim=-1)

    def _decoder_(self, trg, enc, layer_n):
        w = 1.0 + 0.07 * torch.arange(trg.size(1), dtype=torch.float32)
        h = torch.cumsum(self.emb[trg] * w.view(1, -1, 1), dim=1) + enc.mean(dim=1, keepdim=True)[..., :16]
        return self._head(h, layer_n)


class _ToySession:
    """The toy model decoded step-wise, with the interface of model.DecoderSession: the running prefix state of every beam
    follows the ``parent`` rows, as the key / value cache's ancestry does."""

    max_beams = 16

    def __init__(self, model, enc, layer_n):
        self.m, self.n, self.base, self.state, self.s = model, layer_n, enc.mean(dim=1)[..., :16], None, 0

    def step(self, last_tokens, parent=None):
        prev = torch.zeros(1, 16) if self.state is None else self.state
        if self.s > 0:
            prev = prev[parent if parent is not None else torch.arange(last_tokens.numel())]
        self.state = prev + self.m.emb[last_tokens] * (1.0 + 0.07 * self.s)
        self.s += 1
        return self.m._head(self.state + self.base, self.n)


class _ToySessionModel(_ToyDecoderModel):
    def decoder_session(self, enc, layer_n, max_steps):
        return _ToySession(self, enc, layer_n)

    def decoder_session_group(self, encs, layer_ns, max_steps):
        sessions = [_ToySession(self, e, n) for e, n in zip(encs, layer_ns)]

        class Group:
            max_beams = 16

            def step(self, last_tokens, parent=None):
                return torch.stack([s.step(last_tokens[i], None if parent is None else parent[i]) for i, s in enumerate(sessions)])
        return Group()


@pytest.mark.parametrize("min_length,max_length,beam", [(300, 12, 5), (2, 12, 5), (0, 9, 3)])
def test_beam_search_over_decoder_sessions_equals_the_whole_prefix_search(min_length, max_length, beam):
    """The step-wise bookkeeping of BeamInference (``parent`` rows handed to a decoder session, also after EOS has removed
    beams; the exits of an utterance in lockstep) on a toy session model against the whole-prefix search of the same model,
    which the next test pins to the reference's algorithm."""
    from early_exit_transformer_amd.beam import BeamInference
    V, eos = 11, 2
    model = _ToySessionModel(V)
    g = torch.Generator().manual_seed(3)
    encs = [torch.randn(1, 7, 16, generator=g) for _ in range(3)]
    inf = BeamInference()
    kw = dict(vocab_size=V, max_length=max_length, min_length=min_length, SOS_token=1, EOS_token=eos, PAD_token=0, beam_size=beam, pen_alpha=1.0)
    per_exit = []
    for n, enc in enumerate(encs, start=1):
        ta, sa, ba = inf.beam_search(model, enc, n, **kw)
        tb, sb, bb = inf.beam_search(model, enc, n, kv_cache=False, **kw)
        assert ba == bb and [t.tolist() for t in ta] == [t.tolist() for t in tb]
        assert torch.allclose(torch.stack(sa), torch.stack(sb), atol=1e-5)
        per_exit.append((ta, sa, ba))
    together = inf.beam_search_exits(model, encs, [1, 2, 3], **kw)
    if max_length - 1 > min_length:
        assert together is None  # EOS can finalise beams: the exits' beam counts may diverge, no lockstep
    else:
        for (ta, sa, ba), (tb, sb, bb) in zip(per_exit, together):
            assert ba == bb and [t.tolist() for t in ta] == [t.tolist() for t in tb]
            assert torch.allclose(torch.stack(sa), torch.stack(list(sb)), atol=1e-5)


@pytest.mark.parametrize("min_length,max_length,beam", [(300, 12, 5), (2, 12, 5), (0, 9, 3), (4, 6, 4)])
def test_beam_search_equals_the_reference_algorithm(min_length, max_length, beam):
    """early_exit_transformer_amd.beam.BeamInference.beam_search (tensor ops) against the reference's algorithm restated
    with its own Python loops (util/beam_infer.py:198-307; tests/golden/make_golden.py aed_beam_search), including the
    EOS-finalisation branch the reference's defaults never reach (min_length = 300)."""
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    import make_golden as G
    from early_exit_transformer_amd.beam import BeamInference
    V, eos = 11, 2
    model = _ToyDecoderModel(V)
    enc = torch.randn(1, 7, 16, generator=torch.Generator().manual_seed(1))
    inf = BeamInference()
    hit_eos = False
    for layer_n in (1, 3):
        want_t, want_s, want_best, _ = G.aed_beam_search(model, enc, layer_n, max_length, beam_size=beam, alpha=1.0, sos=1, eos=eos,
                                                         V=V, min_length=min_length)
        got_t, got_s, got_best = inf.beam_search(model, enc, layer_n, vocab_size=V, max_length=max_length, min_length=min_length,
                                                 SOS_token=1, EOS_token=eos, PAD_token=0, beam_size=beam, pen_alpha=1.0)
        assert got_best == want_best
        assert len(got_t) == len(want_t)
        for a, b in zip(got_t, want_t):
            assert a.tolist() == b.tolist()
        assert torch.allclose(torch.stack(got_s), torch.stack(want_s), atol=1e-6)
        hit_eos = hit_eos or any(len(t) < max_length + 1 for t in want_t)
    if min_length <= 2:
        assert hit_eos, "the EOS branch must be exercised when min_length allows it"
    if min_length >= max_length:
        assert not hit_eos


def test_named_tensor_index_matches_the_module_tree_walks():
    """model._named_tensors (the cached index the training forward uses) == named_parameters() / state_dict(keep_vars=True): names,
    order and the very tensors -- also after .to() replaced them and after load_state_dict."""
    from conftest import base_kwargs
    from early_exit_transformer_amd.model import Early_conformer, full_conformer, _named_tensors
    kw = base_kwargs(n_enc_exits=2, n_enc_layers=1, d_feed_forward=64)
    fkw = {k: v for k, v in kw.items() if k != "src_pad_idx"}
    for m in (Early_conformer(**kw), full_conformer(trg_pad_idx=126, n_dec_layers=1, **fkw)):
        for _ in range(2):
            a, b = _named_tensors(m)
            want_p = list(m.named_parameters())
            assert [n for n, _ in a] == [n for n, _ in want_p] and all(x is y for (_, x), (_, y) in zip(a, want_p))
            sd = m.state_dict(keep_vars=True)
            assert [n for n, _ in b] == list(sd.keys()) and all(t is sd[n] for n, t in b)
            m = m.double().float()  # replaces every parameter's data and every buffer tensor
            m.load_state_dict(m.state_dict())
