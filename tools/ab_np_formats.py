import os, subprocess, sys, json, statistics
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("abv", "tools/ab_variants.py"); abv = importlib.util.module_from_spec(spec); spec.loader.exec_module(abv)
lib = os.path.abspath("early_exit_transformer_amd/csrc/libeec_npx.so")
res = {}
for r in range(3):
    for ov in ["", "head=3", "head=3,qkv=3", "head=3,qkv=3,glu=3,front=3"]:
        env = dict(os.environ, EEC_LIB_PATH=lib, EEC_NP_OVERRIDE=ov)
        out = subprocess.run([sys.executable, "-c", abv.CHILD], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if line: res.setdefault(ov or "all8", []).append(json.loads(line[-1]))
        else: print("FAILED", ov, out.stderr[-300:])
for k, rs in res.items():
    print(f"{k:32s} fwd {statistics.median(x['fwd_ms'] for x in rs):.3f} chain {statistics.median(x['chain_us'] for x in rs):.1f} glu {statistics.median(x['glu_us'] for x in rs):.1f} attn {statistics.median(x['attn_us'] for x in rs):.1f}")
