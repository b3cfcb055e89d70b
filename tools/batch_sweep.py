"""Forward time of the default model (12 layers, d_model 256, mel [80 x 1027]) against the batch size: one row tile (64
frames) per workgroup means B utterances use 4 B of the 256 CUs, so small batches are latency runs of the same chain."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from early_exit_transformer_amd import synth  # noqa: E402
from early_exit_transformer_amd.model import Early_conformer  # noqa: E402

T = 1027
net = Early_conformer(device="cuda", **bench.CFG).eval()
net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0, style="init"))
net = net.cuda()
for B in (1, 2, 4, 8, 16, 32, 64, 128):
    mel = synth.synth_mel(B, 80, T, seed=0).cuda()
    lens = torch.full((B,), T, dtype=torch.int64)
    with torch.no_grad():
        for _ in range(5):
            net(mel, lens)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            net(mel, lens)
        torch.cuda.synchronize()
    d = (time.perf_counter() - t0) / n
    print(f"B = {B:3d}: {d * 1e3:7.3f} ms per forward, {B * T / d / 1e6:6.2f} M mel-frames/s", flush=True)
