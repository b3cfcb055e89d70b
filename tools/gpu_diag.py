"""GPU bring-up diagnostics: per-sub-step error of the HIP encoder against the CPU oracle.
Run on the GPU box:  python tools/gpu_diag.py [small|default] > gpurun_out/diag.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import conformer_ref as R
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import Early_conformer

which = sys.argv[1] if len(sys.argv) > 1 else "small"
if which == "small":
    kw = dict(src_pad_idx=0, n_enc_exits=2, enc_voc_size=256, dec_voc_size=256, d_model=256, n_head=8, max_len=2000,
              d_feed_forward=256, n_enc_layers=1, features_length=80, drop_prob=0.1, depthwise_kernel_size=31, device="cuda")
    B, T = 3, 131
    lens = torch.tensor([131, 100, 77])
else:
    kw = dict(src_pad_idx=0, n_enc_exits=6, enc_voc_size=256, dec_voc_size=256, d_model=256, n_head=8, max_len=2000,
              d_feed_forward=2048, n_enc_layers=2, features_length=80, drop_prob=0.1, depthwise_kernel_size=31, device="cuda")
    B, T = 4, 1027
    lens = torch.tensor([1027, 903, 771, 642])
ref = R.EarlyConformerRef(**{**kw, "device": "cpu"}).eval()
sd = synth.synth_state_dict(ref.state_dict(), seed=0, style="trained")
ref.load_state_dict(sd)
mel = synth.synth_mel(B, 80, T)
with torch.no_grad():
    steps = R.trace_substeps(ref, mel, lens)
    want = ref(mel, lens)
m = Early_conformer(**kw).eval()
m.load_state_dict(sd)
m = m.cuda()
names = ["stem"] + [f"L{i//4}.{['ffn1','attn','conv','ffn2'][i%4]}" for i in range(len(steps) - 1)]
for prec in ("f16f8", "f16x3", "mixed", "f16"):
    m.precision = prec
    print(f"== precision {prec}")
    with torch.no_grad():
        for k, (nm, s) in enumerate(zip(names, steps)):
            _, _, x = m._run_encoder(mel.cuda(), lens, want_out=False, stop_after=k, want_x=True)
            d = (x.cpu() - s).abs()
            print(f"  step {k:2d} {nm:10s} max|d| {d.max().item():.3e}  mean {d.mean().item():.3e}  ref rms {s.pow(2).mean().sqrt().item():.3f}  nan {torch.isnan(x).any().item()}")
        got = m(mel.cuda(), lens).cpu()
    d = (got - want).abs()
    print(f"  log-probs: max|d| {d.max().item():.3e} mean {d.mean().item():.3e} per-exit {[f'{d[e].max().item():.2e}' for e in range(d.size(0))]}")
    print(f"  argmax mismatch frac {(got.argmax(-1) != want.argmax(-1)).float().mean().item():.2e}")
torch.cuda.synchronize()
for prec in ("f16f8", "f16x3", "f16"):
    m.precision = prec
    melc = mel.cuda()
    with torch.no_grad():
        for _ in range(3): m(melc, lens)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(10): m(melc, lens)
        torch.cuda.synchronize(); dt = (time.time() - t) / 10
    print(f"timing {prec}: {dt*1e3:.3f} ms/forward  B={B} T={T}  {B*T/dt:.3e} mel-frames/s")
