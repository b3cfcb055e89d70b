"""Diagnostic: phase timelines of the qkv / proj_glu / dw_pw2 kernels (needs a library built with -DEEC_TL:
tools/build_variant.sh tl "-DEEC_TL" linear.hip conv.hip; run with EEC_LIB_PATH=.../libeec_tl.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from early_exit_transformer_amd import capi, synth
from early_exit_transformer_amd.model import Early_conformer
prec = sys.argv[1] if len(sys.argv) > 1 else "f16f8"
kw = dict(src_pad_idx=0, n_enc_exits=1, enc_voc_size=256, dec_voc_size=256, d_model=256, n_head=8, max_len=2000,
          d_feed_forward=2048, n_enc_layers=2, features_length=80, drop_prob=0.1, depthwise_kernel_size=31, device="cuda")
m = Early_conformer(**kw).eval(); m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0)); m = m.cuda(); m.precision = prec
mel = synth.synth_mel(64, 80, 1027).cuda(); lens = torch.full((64,), 1027)
lib = capi.load()
with torch.no_grad():
    for _ in range(4): m(mel, lens)
    torch.cuda.synchronize()
names = {"qkv": ["prologue", "barrier", "Q gemm", "Q store+fill", "K gemm", "K store", "V gemm", "V store"],
         "glu": ["loads->lds", "barrier", "out-proj gemm", "tile write", "barrier", "resid+LN+planes", "barrier", "value gemm", "gate gemm", "GLU+store"],
         "attn": ["stage K,V", "barrier", "QK/softmax/PV", "normalise+store"],
         "dw": ["stage g+taps", "barrier", "depthwise+SiLU", "barrier", "pw2 gemm", "residual rmw"]}
for nm, ph in names.items():
    if not hasattr(lib, 'eec_tl_read_' + nm): continue
    fn = getattr(lib, "eec_tl_read_" + nm)
    buf = (C.c_ulonglong * (8 * 2 * 16))()
    fn(buf)
    a = np.array(buf, dtype=np.int64).reshape(8, 2, 16)
    print(f"== {nm} (ticks; phases: {ph})")
    for blk in (0, 5):
        for wv in (0, 1):
            t = a[blk, wv, :len(ph) + 1]
            print(f"  block {blk} wave {'0L'[wv]}: total {t[-1]-t[0]:6d} | " + " ".join(f"{int(v):5d}" for v in np.diff(t)))
            if nm == "glu" and a[blk, wv, 11] > 0:  # fused attention prologue: stamps 11..15 relative to the kernel entry
                print("      attention: Q + first fetch issued / key blocks 0, 1, 2, 3 done at " + " ".join(f"{int(v - a[blk, wv, 0]):6d}" for v in a[blk, wv, 11:16]))
