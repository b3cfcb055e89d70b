#!/bin/bash
# bench each libeec_<v>.so variant (tuning experiments): prints ms/step, FFN us, modes
for v in "$@"; do
  lib=early_exit_transformer_amd/csrc/libeec_$v.so; [ "$v" = base ] && lib=early_exit_transformer_amd/csrc/libeec.so
  EEC_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/var_$v.json")); print("$v", d["ms_per_step"], "fwd", d["forward_only"]["ms_per_step"], {k:(v["avg_us"],v["n"]) for k,v in d["kernel_time"].items() if v["n"]})
except Exception as e: print("$v failed", e)
PY
done
