"""100 cached decoder steps (10 beams, 6 layers, d_model 256, T' = 256) and nothing else: the workload to put under
`rocprofv3 --kernel-trace --stats -- python3 tools/decoder_step_profile.py`; summarise with tools/rocprof_db_summary.py
(profiles/r02_decoder_step_kernel_stats.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from early_exit_transformer_amd import synth  # noqa: E402
from early_exit_transformer_amd.model import full_conformer  # noqa: E402

fc = full_conformer(trg_pad_idx=126, n_dec_layers=6, device="cuda", **{k: v for k, v in bench.CFG.items() if k != "src_pad_idx"}).eval()
fc.load_state_dict(synth.synth_state_dict(fc.state_dict(), seed=4, style="init"))
fc = fc.cuda()
enc = torch.randn(1, 256, 256, device="cuda")
with torch.no_grad():
    sess = fc.decoder_session(enc, 1, 200)
    last = torch.randint(3, 256, (10,), device="cuda")
    parent = torch.randint(0, 10, (10,), device="cuda")
    sess.step(last[:1])
    for _ in range(100):
        sess.step(last, parent)
torch.cuda.synchronize()
