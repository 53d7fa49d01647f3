#!/bin/bash
# lab: variant builds of the bf16 256^2 GEMM -> build_lab/gemm_<TAG>/libdk_hip.so (DK_HIP_LIB selects one)
# usage: TAG=str1 DEFS="-DDK_V3_STR=1 -DDK_V3_PH1=8" scripts/build_gemm_var.sh
set -e
cd "$(dirname "$0")/.."
d=build_lab/gemm_${TAG:-base}
mkdir -p $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${DEFS:-} -c diffusionkit_amd/csrc/gemm256v3.hip -o $d/gemm256v3.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libdk_hip.so $(ls diffusionkit_amd/csrc/build/*.o | grep -v gemm256v3.o) $d/gemm256v3.o
rm $d/gemm256v3.o
