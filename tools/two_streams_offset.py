"""Experiment: two concurrent B=32 forwards on two HIP streams, the second started a fraction of a layer later (the streams then stay
out of phase: when one half of the chip is in a stage boundary the other is in a chunk loop) against one B=64 forward."""
import sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import Early_conformer
import bench
m = Early_conformer(**bench.CFG, device="cuda").eval(); m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0)); m = m.cuda()
m2 = Early_conformer(**bench.CFG, device="cuda").eval(); m2.load_state_dict(m.state_dict()); m2 = m2.cuda()
mel = synth.synth_mel(64, 80, 1027).cuda(); lens = torch.full((64,), 1027)
ma, mb = mel[:32].contiguous(), mel[32:].contiguous(); la, lb = lens[:32], lens[32:]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 40
with torch.no_grad():
    for _ in range(5): m(mel, lens); m(ma, la); m2(mb, lb)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N): m(mel, lens)
    torch.cuda.synchronize(); print(f"B=64 one stream: {(time.perf_counter() - t) / N * 1e3:.3f} ms per 64 utterances")
    for off_us in (0, 40, 80, 120, 160, 200):
        torch.cuda.synchronize(); t = time.perf_counter()
        with torch.cuda.stream(s2):
            if off_us: torch.cuda._sleep(int(off_us * 2100))
        for _ in range(N):
            with torch.cuda.stream(s1): m(ma, la)
            with torch.cuda.stream(s2): m2(mb, lb)
        torch.cuda.synchronize()
        print(f"2 x B=32, second stream {off_us:3d} us behind: {((time.perf_counter() - t) * 1e3 - off_us * 1e-3) / N:.3f} ms per 64 utterances")
