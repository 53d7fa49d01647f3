#!/bin/bash
# lab: ablation builds of the halo conv kernel -> build_lab/halo<n><TAG>/libdk_hip.so (DK_HIP_LIB selects one);
# usage: ABL="1 2 4 ..." [DEFS="-DCH_SGB=0" TAG=_nosgb] scripts/build_halo_abl.sh
set -e
cd "$(dirname "$0")/.."
for n in ${ABL:-0}; do
  mkdir -p build_lab/halo$n${TAG:-}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCH_ABL=$n ${DEFS:-} -c diffusionkit_amd/csrc/conv_halo.hip -o build_lab/halo$n${TAG:-}/conv_halo.o &
done
wait
for n in ${ABL:-0}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_lab/halo$n${TAG:-}/libdk_hip.so $(ls diffusionkit_amd/csrc/build/*.o | grep -v conv_halo.o) build_lab/halo$n${TAG:-}/conv_halo.o
  rm build_lab/halo$n${TAG:-}/conv_halo.o
done
