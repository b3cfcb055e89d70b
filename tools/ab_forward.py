import sys, time, torch, os
sys.path.insert(0, os.getcwd())
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import Early_conformer
import bench
m = Early_conformer(**bench.CFG, device="cuda").eval(); m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0)); m = m.cuda()
mel = synth.synth_mel(64, 80, 1027).cuda(); lens = torch.full((64,), 1027)
with torch.no_grad():
    for _ in range(5): m(mel, lens)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(30): m(mel, lens)
    torch.cuda.synchronize(); print(os.environ.get("EEC_LIB_PATH","base").split("/")[-1], "fwd ms", (time.perf_counter()-t)/30*1e3)
