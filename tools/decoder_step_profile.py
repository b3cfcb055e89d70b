"""100 cached decoder steps (10 beams, 6 layers, d_model 256, T' = 256; `python3 tools/decoder_step_profile.py 6`: six exits
in lockstep) and nothing else: the workload to put under
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
with torch.no_grad():
    last = torch.randint(3, 256, (10,), device="cuda")
    parent = torch.randint(0, 10, (10,), device="cuda")
    if n == 1:
        sess = fc.decoder_session(enc, 1, 200)
        sess.step(last[:1])
        for _ in range(100):
            sess.step(last, parent)
    else:
        group = fc.decoder_session_group([enc] * n, list(range(1, n + 1)), 200)
        group.step(last[:1].repeat(n, 1))
        for _ in range(100):
            group.step(last.repeat(n, 1), parent.repeat(n, 1))
torch.cuda.synchronize()
