"""Per-kernel-family times of the BASELINE configs[2] geometry (18 layers, d_model 512, 8 heads x 64) forward, B = 64, T = 1027, default mode:
mean launch time by family from the library's own HIP-event profile (model.set_profiling)."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import Early_conformer
cfg = dict(bench.CFG, d_model=512, n_enc_layers=3)
m = Early_conformer(**cfg, device="cuda").eval(); m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0)); m = m.cuda()
mel = synth.synth_mel(64, 80, 1027).cuda(); lens = torch.full((64,), 1027)
with torch.no_grad():
    for _ in range(3): m(mel, lens)
    torch.cuda.synchronize(); ts = []
    for _ in range(10):
        t = time.perf_counter(); m(mel, lens); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    m.set_profiling(True)
    for _ in range(3): m(mel, lens)
    torch.cuda.synchronize()
    prof = m.read_profile()
ts.sort()
print(json.dumps({"fwd_ms": ts[len(ts) // 2] * 1e3, "per_launch_us": {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items() if v[1]},
                  "launches_per_forward": {k: v[1] // 3 for k, v in prof.items() if v[1]}}))
