#!/bin/bash
# build_variant.sh <name> "<defines>" <file.hip> [file.hip ...]
# Recompiles the named sources with extra -D flags and links libeec_<name>.so against the base objects
# (tuning experiments; bench them with tools/ab_variants.py / tools/ab_train.py).  ffn.hip is built with -DEEC_CHAIN_MINIMAL (only the
# three chain-kernel variants of the production plan in the default f16x3 mode, d_model 256; add -DEEC_CHAIN_MINIMAL_F8 to the defines for
# the f16f8 plan) unless EEC_FULL=1.  The d_model 512 and the training objects (ffn512 / ffn_train / ffn_train_bwd) are linked as built.
set -e
name=$1; defs=$2; shift 2
cd "$(dirname "$0")/../early_exit_transformer_amd/csrc"
[ -f build/capi.o ] || make -s -j6
objs=""
for s in capi ffn linear attention conv stem ctc pack frontend ctc_beam train_kernels train_attention train decoder decoder_step decoder_train; do
  if [[ " $* " == *" $s.hip "* ]]; then
    extra=""
    if [ "$s" = ffn ] && [ "${EEC_FULL:-0}" != 1 ]; then extra="-DEEC_CHAIN_MINIMAL"; fi
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed $defs $extra -c $s.hip -o build/${s}_$name.o
    objs="$objs build/${s}_$name.o"
  else
    objs="$objs build/$s.o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/ffn512.o build/ffn_train.o build/ffn_train_bwd.o -o libeec_$name.so
echo built libeec_$name.so
