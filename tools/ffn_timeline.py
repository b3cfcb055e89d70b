"""Diagnostic: per-slot cycle timeline of the FFN kernel (needs libeec_tl.so built with -DEEC_TIMELINE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from early_exit_transformer_amd import capi, synth
from early_exit_transformer_amd.model import Early_conformer
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
kw = dict(src_pad_idx=0, n_enc_exits=1, enc_voc_size=256, dec_voc_size=256, d_model=256, n_head=8, max_len=2000,
          d_feed_forward=2048, n_enc_layers=2, features_length=80, drop_prob=0.1, depthwise_kernel_size=31, device="cuda")
m = Early_conformer(**kw).eval(); m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0)); m = m.cuda(); m.precision = prec
mel = synth.synth_mel(64, 80, 1027).cuda(); lens = torch.full((64,), 1027)
lib = capi.load()
with torch.no_grad():
    for _ in range(3): m(mel, lens)
    torch.cuda.synchronize()
    lib.eec_debug_timeline(None, 0)          # arm
    m(mel, lens)   # 2 layers: [ffn1->qkv], [dw->ffn2->ffn1->qkv], [dw->ffn2]; build with -DEEC_TL_NS=2 to keep the middle one
    torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 2 * 128))()
lib.eec_debug_timeline(buf, 8 * 2 * 128)
a = np.array(buf, dtype=np.int64).reshape(8, 2, 128)
for blk in (0, 3):
    for role, nm in ((0, "producer w0"), (1, "consumer w4")):
        t = a[blk, role]; n = int((t > 0).sum()); t = t[:n] - a[blk, 0, 0]
        print(f"block {blk} {nm}: {n} stamps; total {t[-1]} cycles (100MHz ticks? see deltas)")
        print("   ", " ".join(str(int(v)) for v in np.diff(t)))

if hasattr(lib, "eec_debug_ksteps"):
    ks = (C.c_ulonglong * 256)()
    lib.eec_debug_ksteps(ks)            # drop whatever the warm-ups recorded, reset index
    with torch.no_grad():
        m._run_encoder(mel, lens, want_out=False, stop_after=1)
    lib.eec_debug_ksteps(ks)
    k = np.array(ks, dtype=np.int64); k = k[k > 0]
    print("consumer wave 0 of block 0: cycles between consecutive k-step ends (8 per slot):")
    d = np.diff(k)
    for i in range(0, min(len(d), 72), 8): print("   ", d[i:i+8].tolist())
