#!/bin/bash
# lab: ablation builds of an attention kernel -> build_lab/attn<K><n><TAG>/libdk_hip.so (DK_HIP_LIB selects one);
# usage: [K=3|4] ABL="1 2 4 ..." [DEFS="-DDK4_PRIO=1" TAG=_prio] scripts/build_attn_abl.sh   (K: attention3.hip / attention4.hip)
set -e
cd "$(dirname "$0")/.."
K=${K:-3}
for n in ${ABL:-0}; do
  mkdir -p build_lab/attn$K$n${TAG:-}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-honor-nans -fno-slp-vectorize -DDK${K}_ABL=$n ${DEFS:-} -c diffusionkit_amd/csrc/attention$K.hip -o build_lab/attn$K$n${TAG:-}/attention$K.o &
done
wait
for n in ${ABL:-0}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_lab/attn$K$n${TAG:-}/libdk_hip.so $(ls diffusionkit_amd/csrc/build/*.o | grep -v attention$K.o) build_lab/attn$K$n${TAG:-}/attention$K.o
  rm build_lab/attn$K$n${TAG:-}/attention$K.o
done
