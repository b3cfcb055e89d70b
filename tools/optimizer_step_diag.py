#!/usr/bin/env python3
"""Diagnostic: wall time of the training step (train.py:53-70) by gradient placement (fresh tensors per step / flat buckets of
enable_data_parallel) and optimizer implementation (torch AdamW foreach / fused), and the GPU time of clip + AdamW alone."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import Early_conformer, exit_ctc_losses
B, T = 64, 1027
mel = synth.synth_mel(B, 80, T, seed=0).cuda(); lengths = torch.full((B,), T, dtype=torch.int64)
tgt, tl = synth.synth_targets(B, 40, 256, seed=0); tgt, tl = tgt.cuda(), tl.cuda()
for mode in ("plain", "dp"):
    for fused in (False, True):
        tm = Early_conformer(device="cuda", **bench.CFG)
        tm.load_state_dict(synth.synth_state_dict(tm.state_dict(), seed=2, style="init"))
        tm = tm.cuda().train(); tm.train_passes = 3
        params = list(tm.parameters())
        if mode == "dp": tm.enable_data_parallel(B)
        opt = torch.optim.AdamW(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.1, **({'fused': True} if fused else {}))  # fused=False would also switch foreach off
        ev = []
        def step(timed=False):
            opt.zero_grad(set_to_none=True)
            loss = exit_ctc_losses(tm(mel, lengths), tgt, tl).sum(); loss.backward()
            if mode == "dp": tm.sync_gradients()
            if timed:
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(); torch.nn.utils.clip_grad_norm_(params, 1.0); e[1].record(); opt.step(); e[2].record(); ev.append(e)
            else:
                torch.nn.utils.clip_grad_norm_(params, 1.0); opt.step()
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
        for _ in range(3): step(True)
        torch.cuda.synchronize()
        print(f"gradients {mode:5s} AdamW {'fused  ' if fused else 'foreach'}: {dt * 1e3:6.2f} ms per step; GPU time of clip {ev[-1][0].elapsed_time(ev[-1][1]):.2f} ms, AdamW {ev[-1][1].elapsed_time(ev[-1][2]):.2f} ms", flush=True)
        del tm, opt, params
        torch.cuda.empty_cache()
