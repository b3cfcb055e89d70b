"""Host side of the training step (bench.py train_step loop without the all-reduce): wall time per step against the time the
host needs to ISSUE a step (no synchronisation inside the loop), split by phase.  Under rocprofv3, tools/gpu_busy_union.py
shows where the GPU waits for the host."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from early_exit_transformer_amd import synth, parallel
from early_exit_transformer_amd.model import Early_conformer, exit_ctc_losses
B, T = 64, 1027
tm = Early_conformer(device="cuda", **bench.CFG)
tm.load_state_dict(synth.synth_state_dict(tm.state_dict(), seed=2, style="init"))
tm = tm.cuda().train(); tm.train_passes = 3
params = list(tm.parameters())
opt = torch.optim.AdamW(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.1, **({'fused': True} if int(os.environ.get('EEC_FUSED_ADAMW', '0')) else {}))
mel = synth.synth_mel(B, 80, T, seed=0).cuda(); lengths = torch.full((B,), T, dtype=torch.int64)
tgt, tl = synth.synth_targets(B, 40, 256, seed=0); tgt, tl = tgt.cuda(), tl.cuda()
def step(marks=None):
    t = time.perf_counter()
    opt.zero_grad(set_to_none=True); a = time.perf_counter()
    out = tm(mel, lengths); b = time.perf_counter()
    loss = exit_ctc_losses(out, tgt, tl).sum(); c = time.perf_counter()
    loss.backward(); d = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(params, 1.0); e = time.perf_counter()
    opt.step(); f = time.perf_counter()
    if marks is not None: marks.append([a-t, b-a, c-b, d-c, e-d, f-e])
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
marks = []
t0 = time.perf_counter()
for _ in range(10): step(marks)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
import numpy as np
m = np.array(marks[2:]).mean(0) * 1e3
print("per step: total %.2f ms, host issue %.2f ms; host ms: zero_grad %.2f forward %.2f loss %.2f backward %.2f clip %.2f adamw %.2f" % (t_all/10*1e3, t_issue/10*1e3, *m))
