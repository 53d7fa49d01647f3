#!/bin/bash
# lab: libdk_hip.so with ONE kernel file's device code built under another LLVM machine scheduler
#   scripts/build_misched.sh attention3 gcn-max-ilp "-fno-honor-nans -fno-slp-vectorize"  -> build_lab/<file>_<sched>/libdk_hip.so
# (-mllvm options reach the host compilation too and the x86 backend rejects the GCN schedulers: device and host passes are run
#  separately and the device code object is embedded by hand)
set -e
cd "$(dirname "$0")/.."
F=$1; S=$2; X=${3:-}
D=build_lab/${F}_$S
mkdir -p $D
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $X"
/opt/rocm/bin/hipcc $FL -mllvm -misched=$S --offload-device-only -c diffusionkit_amd/csrc/$F.hip -o $D/dev.hipfb  # (already an offload bundle)
/opt/rocm/bin/hipcc $FL --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang $D/dev.hipfb -c diffusionkit_amd/csrc/$F.hip -o $D/$F.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libdk_hip.so $(ls diffusionkit_amd/csrc/build/*.o | grep -v "/$F.o") $D/$F.o
rm -f $D/dev.hipfb $D/$F.o
ls -la $D/libdk_hip.so
