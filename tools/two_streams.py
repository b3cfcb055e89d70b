"""Experiment: one B=64 forward vs two concurrent B=32 forwards on two HIP streams (same weights)."""
import sys, time, torch, os
sys.path.insert(0, os.getcwd())
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import Early_conformer
import bench
m = Early_conformer(**bench.CFG, device="cuda").eval(); m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0)); m = m.cuda()
m2 = Early_conformer(**bench.CFG, device="cuda").eval(); m2.load_state_dict(m.state_dict()); m2 = m2.cuda()
mel = synth.synth_mel(64, 80, 1027).cuda(); lens = torch.full((64,), 1027)
ma, mb = mel[:32].contiguous(), mel[32:].contiguous(); la, lb = lens[:32], lens[32:]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one(n):
    for _ in range(n): m(mel, lens)
def two(n):
    for _ in range(n):
        with torch.cuda.stream(s1): m(ma, la)
        with torch.cuda.stream(s2): m2(mb, lb)
def seq(n):
    for _ in range(n):
        m(ma, la); m2(mb, lb)
def two64(n):
    for _ in range(n // 2):
        with torch.cuda.stream(s1): m(mel, lens)
        with torch.cuda.stream(s2): m2(mel, lens)
with torch.no_grad():
    for f, nm in ((two64, "B=64 x 2 streams (per 64)"), (one, "B=64 one stream"), (seq, "2 x B=32 same stream"), (two, "2 x B=32 two streams"), (one, "B=64 one stream")):
        f(5); torch.cuda.synchronize(); t = time.perf_counter(); f(30); torch.cuda.synchronize()
        print(f"{nm:28s} {(time.perf_counter()-t)/30*1e3:.3f} ms per 64 utterances")
