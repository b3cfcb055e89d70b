"""cProfile of the host side of the training step (bench.py's loop): where the Python time of a step goes."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.getcwd())
import torch
import bench
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import Early_conformer, exit_ctc_losses
m = Early_conformer(device="cuda", **bench.CFG)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=2, style="init"))
m = m.cuda().train(); m.train_passes = 3
params = list(m.parameters())
opt = torch.optim.AdamW(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.1, fused=True)
mel = synth.synth_mel(64, 80, 1027, seed=0).cuda(); lens = torch.full((64,), 1027)
tgt, tl = synth.synth_targets(64, 42, 256, seed=0); tgt, tl = tgt.cuda(), tl.cuda()
def step():
    opt.zero_grad(set_to_none=True)
    loss = exit_ctc_losses(m(mel, lens), tgt, tl).sum(); loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:4500])
