"""Diagnostic: start/end ticks of the first 1024 attention workgroups (library built with -DEEC_TL)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from early_exit_transformer_amd import capi, synth
from early_exit_transformer_amd.model import Early_conformer
kw = dict(src_pad_idx=0, n_enc_exits=1, enc_voc_size=256, dec_voc_size=256, d_model=256, n_head=8, max_len=2000,
          d_feed_forward=2048, n_enc_layers=1, features_length=80, drop_prob=0.1, depthwise_kernel_size=31, device="cuda")
m = Early_conformer(**kw).eval(); m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0)); m = m.cuda()
mel = synth.synth_mel(64, 80, 1027).cuda(); lens = torch.full((64,), 1027)
lib = capi.load()
with torch.no_grad():
    for _ in range(4): m(mel, lens)
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 2048)()
lib.eec_tl_read_attn_all(buf)
a = np.array(buf, dtype=np.int64).reshape(1024, 2)
# s_memtime is per XCD and the block -> XCD map is not a pure modulo: cluster the start stamps instead
st_all = a[:, 0]
order = np.argsort(st_all)
clusters, cur = [], [order[0]]
for i in order[1:]:
    if st_all[i] - st_all[cur[-1]] > 200000:
        clusters.append(cur); cur = []
    cur.append(i)
clusters.append(cur)
for c in clusters:
    g = a[np.array(c)]
    t0 = g[:, 0].min()
    st, en = g[:, 0] - t0, g[:, 1] - t0
    print(f"cluster of {len(c):4d} WGs: span {en.max():6d} ticks; WG lifetime mean {(en - st).mean():.0f}; "
          f"start pct {np.percentile(st, [0, 25, 50, 75, 100]).astype(int).tolist()}")
