import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import full_conformer
fc = full_conformer(trg_pad_idx=126, n_dec_layers=6, device="cuda", **{k: v for k, v in bench.CFG.items() if k != "src_pad_idx"}).eval()
fc.load_state_dict(synth.synth_state_dict(fc.state_dict(), seed=4, style="init")); fc = fc.cuda()
enc = torch.randn(1, 256, 256, device="cuda")
with torch.no_grad():
    sess = fc.decoder_session(enc, 1, 200)
    last = torch.randint(3, 256, (10,), device="cuda"); par = torch.randint(0, 10, (10,), device="cuda")
    sess.step(last[:1])
    for _ in range(100): sess.step(last, par)
torch.cuda.synchronize()
