import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import full_conformer
fc = full_conformer(trg_pad_idx=126, n_dec_layers=6, device="cuda", **{k: v for k, v in bench.CFG.items() if k != "src_pad_idx"}).eval()
fc.load_state_dict(synth.synth_state_dict(fc.state_dict(), seed=4, style="init")); fc = fc.cuda()
enc = torch.randn(10, 256, 256, device="cuda"); 
for S in (1, 40, 85):
    tok = torch.randint(3, 256, (10, S), device="cuda")
    with torch.no_grad():
        for _ in range(5): fc._decoder_(tok, enc, 1)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): fc._decoder_(tok, enc, 1)
        torch.cuda.synchronize(); d = (time.perf_counter() - t) / 50
        t = time.perf_counter()
        for _ in range(50): fc._decoder_(tok, enc, 1)
        dc = (time.perf_counter() - t) / 50
        torch.cuda.synchronize()
    print(f"S={S}: {d*1e3:.3f} ms per _decoder_ call (sync'd average); CPU-side issue time {dc*1e3:.3f} ms")
