#!/usr/bin/env python3
"""Same-box A/B of library builds on the training step: interleaved rounds, one subprocess per (round, variant).
    python tools/ab_train.py [--rounds 3] base=.../libeec.so v1=.../libeec_v1.so[,ENV=VAL ...] ..."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]; rounds = 3
if args and args[0] == "--rounds": rounds = int(args[1]); args = args[2:]
variants = [a.split("=", 1) for a in args]; res = {n: [] for n, _ in variants}
for r in range(rounds):
    for n, path in variants:
        path, *extra = path.split(",")  # lib[,ENV=VAL ...]
        env = dict(os.environ, EEC_LIB_PATH=os.path.abspath(path), **dict(e.split("=", 1) for e in extra))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_step_time.py")], cwd=ROOT, env=env, capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if line:
            rec = json.loads(line[-1]); res[n].append(rec["train_ms"])
            print(n, " ".join(f"{k}={v:.3f}" for k, v in rec.items()), flush=True)
        else: print(n, "FAILED", out.stderr[-300:])
for n, v in res.items():
    if v: print(f"{n:12s} train step ms median {statistics.median(v):.3f} min {min(v):.3f}")
