#!/usr/bin/env python3
"""A few training steps of a BASELINE config on the HIP training kernels, for rocprofv3 --kernel-trace --stats:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -- python3 tools/train_profile.py [--d-model 256] [--layers 2] [--passes 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from early_exit_transformer_amd import synth  # noqa: E402
from early_exit_transformer_amd.model import Early_conformer, exit_ctc_losses  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d-model", type=int, default=256)
ap.add_argument("--layers", type=int, default=2)
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
cfg = dict(bench.CFG, d_model=a.d_model, n_enc_layers=a.layers)
m = Early_conformer(device="cuda", **cfg)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=2, style="init"))
m = m.cuda().train()
m.train_passes = a.passes
mel = synth.synth_mel(a.batch, 80, 1027, seed=0).cuda()
lens = torch.full((a.batch,), 1027)
tgt, tl = synth.synth_targets(a.batch, 42, 256, seed=0)
tgt, tl = tgt.cuda(), tl.cuda()
for i in range(a.steps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.zero_grad(set_to_none=True)
    out = m(mel, lens)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss = exit_ctc_losses(out, tgt, tl).sum()
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"step {i}: forward {1e3 * (t1 - t0):.2f} ms, loss + backward {1e3 * (t2 - t1):.2f} ms, loss {loss.item():.4f}", flush=True)
