"""Error-budget experiment (library built with -DEEC_NP_EXPERIMENT): max |delta log-prob| of the default 12-layer
model against the CPU oracle when single GEMM groups of the production plan run with 1-pass fp16 operands
(EEC_NP_OVERRIDE), everything else as in the f16f8 default.  Several seeds; B=4 ragged, T=1027."""
import os, sys, subprocess, json
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, HERE)
    import torch
    from oracle import conformer_ref as R
    from early_exit_transformer_amd import synth
    from early_exit_transformer_amd.model import Early_conformer
    kw = dict(src_pad_idx=0, n_enc_exits=6, enc_voc_size=256, dec_voc_size=256, d_model=256, n_head=8, max_len=2000,
              d_feed_forward=2048, n_enc_layers=2, features_length=80, drop_prob=0.1, depthwise_kernel_size=31)
    errs = []
    for seed in (0, 1, 2):
        ref = R.EarlyConformerRef(**{**kw, "device": "cpu"}).eval()
        sd = synth.synth_state_dict(ref.state_dict(), seed=seed, style="trained")
        ref.load_state_dict(sd)
        mel = synth.synth_mel(4, 80, 1027, seed=seed); lens = torch.tensor([1027, 903, 771, 642])
        cache = f"/tmp/np_budget_ref_{seed}.pt"
        if os.path.exists(cache):
            want = torch.load(cache)
        else:
            with torch.no_grad(): want = ref(mel, lens)
            torch.save(want, cache)
        m = Early_conformer(**{**kw, "device": "cuda"}).eval(); m.load_state_dict(sd); m = m.cuda()
        with torch.no_grad(): got = m(mel.cuda(), lens).cpu()
        errs.append((got - want).abs().max().item())
    print("RESULT " + json.dumps(errs))
    sys.exit(0)
DEFAULT = ["", "qkv=1", "att=1", "glu=1", "front=1", "head=1", "qkv=1,att=1", "qkv=1,front=1", "qkv=1,att=1,front=1",
           "qkv=1,att=1,glu=1,front=1", "qkv=1,att=1,glu=1,front=1,head=1"]
for ov in (sys.argv[1:] or DEFAULT):  # e.g. "head=3" "head=3,qkv=3": formats (1, 3, 8) per GEMM group
    ov = "" if ov == "default" else ov
    env = dict(os.environ, EEC_NP_OVERRIDE=ov)
    out = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True).stdout
    res = [l for l in out.splitlines() if l.startswith("RESULT")]
    print(f"{ov or '(default f16f8)':40s} {res[0][7:] if res else 'FAILED'}", flush=True)
