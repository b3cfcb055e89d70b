"""Time of one training step (train.py:53-70 on the HIP training kernels) at BASELINE configs[3]'s per-rank shape: default model,
B = 64 x 1027 mel frames, dropout 0.1, bf16x3; forward + loss + backward + clip + fused AdamW.  Prints one JSON line (used by
tools/ab_train.py for same-box A/B of library builds through EEC_LIB_PATH)."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import Early_conformer, exit_ctc_losses
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m = Early_conformer(device="cuda", **bench.CFG)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=2, style="init"))
m = m.cuda().train(); m.train_passes = passes
params = list(m.parameters())
opt = torch.optim.AdamW(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.1, fused=True)
mel = synth.synth_mel(64, 80, 1027, seed=0).cuda(); lens = torch.full((64,), 1027)
tgt, tl = synth.synth_targets(64, 42, 256, seed=0); tgt, tl = tgt.cuda(), tl.cuda()
def step():
    opt.zero_grad(set_to_none=True)
    loss = exit_ctc_losses(m(mel, lens), tgt, tl).sum(); loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); ts = []
for _ in range(3):
    t = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 5 * 1e3)
# phases by device events (each phase synchronised: the sum is a little above the pipelined step)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
ph = []
for _ in range(5):
    opt.zero_grad(set_to_none=True); ev[0].record()
    out = m(mel, lens); ev[1].record()
    loss = exit_ctc_losses(out, tgt, tl).sum(); ev[2].record()
    loss.backward(); ev[3].record()
    torch.nn.utils.clip_grad_norm_(params, 1.0); opt.step(); ev[4].record()
    torch.cuda.synchronize()
    ph.append([ev[i].elapsed_time(ev[i + 1]) for i in range(4)])
ph = [sorted(c)[2] for c in zip(*ph)]
print(json.dumps({"train_ms": sorted(ts)[1], "min_ms": min(ts), "forward_ms": ph[0], "loss_ms": ph[1], "backward_ms": ph[2], "clip_adamw_ms": ph[3]}))
