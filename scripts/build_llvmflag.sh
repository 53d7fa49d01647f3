#!/bin/bash
# lab: libdk_hip.so with ONE kernel file's device code built with extra -mllvm options (device pass only, see build_misched.sh)
#   scripts/build_llvmflag.sh attention3 tag "-mllvm -amdgpu-schedule-metric-bias=0" "-fno-honor-nans -fno-slp-vectorize"
set -e
cd "$(dirname "$0")/.."
F=$1; T=$2; L=$3; X=${4:-}
D=build_lab/${F}_$T
mkdir -p $D
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $X"
/opt/rocm/bin/hipcc $FL $L --offload-device-only -c diffusionkit_amd/csrc/$F.hip -o $D/dev.hipfb
/opt/rocm/bin/hipcc $FL --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang $D/dev.hipfb -c diffusionkit_amd/csrc/$F.hip -o $D/$F.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libdk_hip.so $(ls diffusionkit_amd/csrc/build/*.o | grep -v "/$F.o") $D/$F.o
rm -f $D/dev.hipfb $D/$F.o
