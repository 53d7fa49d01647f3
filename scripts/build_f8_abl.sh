#!/bin/bash
# lab builds of libdk_hip.so with fp8-GEMM ablations: scripts/build_f8_abl.sh "1 2 4 7" -> build_lab/f8abl<n>/libdk_hip.so
set -e
cd "$(dirname "$0")/.."
make -C diffusionkit_amd/csrc -j8 > /dev/null
for n in $1; do
  mkdir -p build_lab/f8abl$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDK_F8_ABL=$n -c diffusionkit_amd/csrc/gemm256f8.hip -o build_lab/f8abl$n/gemm256f8.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_lab/f8abl$n/libdk_hip.so $(ls diffusionkit_amd/csrc/build/*.o | grep -v gemm256f8.o) build_lab/f8abl$n/gemm256f8.o
  rm build_lab/f8abl$n/gemm256f8.o
done
