"""AED decoder call timing on one GPU: `_decoder_` on whole prefixes (eec_decoder_forward) against one cached step
(eec_decoder_step) at 10 beams, and the beam search of bench.py's aed_decode line both ways."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from early_exit_transformer_amd import synth  # noqa: E402
from early_exit_transformer_amd.beam import BeamInference  # noqa: E402
from early_exit_transformer_amd.model import full_conformer  # noqa: E402

fc = full_conformer(trg_pad_idx=126, n_dec_layers=6, device="cuda", **{k: v for k, v in bench.CFG.items() if k != "src_pad_idx"}).eval()
fc.load_state_dict(synth.synth_state_dict(fc.state_dict(), seed=4, style="init"))
fc = fc.cuda()
enc = torch.randn(1, 256, 256, device="cuda")


def avg(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    issued = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n, issued


with torch.no_grad():
    for S in (1, 40, 85):
        tok = torch.randint(3, 256, (10, S), device="cuda")
        d, dc = avg(lambda: fc._decoder_(tok, enc.expand(10, -1, -1), 1))
        print(f"whole prefix S={S}: {d * 1e3:.3f} ms per _decoder_ call; CPU-side issue time {dc * 1e3:.3f} ms")
    sess = fc.decoder_session(enc, 1, 2000)
    last = torch.randint(3, 256, (10,), device="cuda")
    par = torch.randint(0, 10, (10,), device="cuda")
    sess.step(last[:1])
    for mark in (1, 40, 85, 400):
        while sess.s < mark:
            sess.step(last, par)
        d, dc = avg(lambda: sess.step(last, par), n=20)
        print(f"cached step at position {sess.s - 10}..{sess.s}: {d * 1e3:.3f} ms per step; CPU-side issue time {dc * 1e3:.3f} ms")
    encs = [torch.randn(1, 256, 256, device="cuda") for _ in range(6)]
    group = fc.decoder_session_group(encs, list(range(1, 7)), 2000)
    last6, par6 = last.repeat(6, 1), par.repeat(6, 1)
    group.step(last6[:, :1])
    while group.s < 40:
        group.step(last6, par6)
    d, dc = avg(lambda: group.step(last6, par6), n=20)
    print(f"cached step of 6 exits in lockstep at position {group.s - 10}..{group.s}: {d * 1e3:.3f} ms per step of all six; CPU-side issue time {dc * 1e3:.3f} ms")
    inf = BeamInference()
    kw = dict(vocab_size=256, SOS_token=1, EOS_token=2, PAD_token=126, pen_alpha=1.0, beam_size=10, max_length=85)
    for cached in (True, False):
        d, _ = avg(lambda: inf.beam_search(fc, enc, 1, kv_cache=cached, **kw), n=3)
        print(f"beam search, 85 steps, kv_cache={cached}: {d * 1e3:.1f} ms = {d / 85 * 1e3:.3f} ms per step")
    d, _ = avg(lambda: inf.beam_search_exits(fc, encs, list(range(1, 7)), **kw), n=3)
    print(f"beam search of 6 exits in lockstep, 85 steps: {d * 1e3:.1f} ms = {d / 85 * 1e3:.3f} ms per step of all six")
