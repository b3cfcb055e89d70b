#!/usr/bin/env python3
"""A/B of library builds on the GPU box: interleaved rounds, one subprocess per (round, variant).

    python tools/ab_variants.py [--rounds 3] base=early_exit_transformer_amd/csrc/libeec.so v1=.../libeec_v1.so[,ENV=VAL] ...

Each run times the default-config forward (B = 64, T = 1027, f16f8; median of 30 after 5 warm-ups) and the mean
chain-kernel launch (HIP events).  Prints per-variant median / min over the rounds."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch, os, json
sys.path.insert(0, os.getcwd())
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import Early_conformer
import bench
m = Early_conformer(**bench.CFG, device="cuda").eval(); m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0)); m = m.cuda()
mel = synth.synth_mel(64, 80, 1027).cuda(); lens = torch.full((64,), 1027)
ts = []
with torch.no_grad():
    for _ in range(5): m(mel, lens)
    torch.cuda.synchronize()
    for _ in range(30):
        t = time.perf_counter(); m(mel, lens); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    m.set_profiling(True)
    for _ in range(5): m(mel, lens)
    torch.cuda.synchronize()
    prof = m.read_profile()
ts.sort()
print(json.dumps({"fwd_ms": ts[len(ts) // 2] * 1e3, "chain_us": prof["chain"][0] / max(prof["chain"][1], 1) * 1e3,
                  "glu_us": prof["proj_glu"][0] / max(prof["proj_glu"][1], 1) * 1e3, "attn_us": prof["attn"][0] / max(prof["attn"][1], 1) * 1e3,
                  "other_us": {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items() if k not in ("chain", "proj_glu", "attn") and v[1]}}))
'''


def main():
    args = sys.argv[1:]
    rounds = 3
    if args and args[0] == "--rounds":
        rounds = int(args[1])
        args = args[2:]
    variants = [a.split("=", 1) for a in args]
    res = {n: [] for n, _ in variants}
    for r in range(rounds):
        for n, path in variants:
            path, *extra = path.split(",")  # name=path[,KEY=VAL ...]: extra environment for this variant
            env = dict(os.environ, EEC_LIB_PATH=os.path.abspath(path))
            env.update(dict(kv.split("=", 1) for kv in extra))
            out = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(n, "FAILED", out.stderr[-400:])
                continue
            res[n].append(json.loads(line[-1]))
    for n, rs in res.items():
        if rs:
            f = [x["fwd_ms"] for x in rs]
            c = [x["chain_us"] for x in rs]
            print(f"{n:12s} fwd ms median {statistics.median(f):.3f} min {min(f):.3f} | chain us median {statistics.median(c):.1f} min {min(c):.1f}"
                  f" | glu {statistics.median([x['glu_us'] for x in rs]):.1f} attn {statistics.median([x['attn_us'] for x in rs]):.1f}"
                  f" | other {rs[-1].get('other_us')}", flush=True)


if __name__ == "__main__":
    main()
