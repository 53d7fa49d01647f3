#!/bin/bash
# builds build_lab/libdk_hip.so with -DDK_LAB_ABLATIONS (extra kernel instantiations for ablation runs) and the lab binary
set -e
cd "$(dirname "$0")/.."
mkdir -p build_lab/obj
for f in gemm gemm256v3 gemm256v4 gemm256f8 fp8_ops attention attention2 attention4 attention512 elementwise vae_ops conv_halo text_ops profile engine; do
  EXTRA=""; [ "$f" = attention2 ] && EXTRA="-fno-honor-nans"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDK_LAB_ABLATIONS $EXTRA -c diffusionkit_amd/csrc/$f.hip -o build_lab/obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_lab/libdk_hip.so build_lab/obj/*.o
/opt/rocm/bin/hipcc -O2 -std=c++17 -w scripts/gemm_lab.cpp -Iinclude -Lbuild_lab -ldk_hip -Wl,-rpath,'$ORIGIN' -o build_lab/gemm_lab
/opt/rocm/bin/hipcc -O2 -std=c++17 -w scripts/attn_lab.cpp -Iinclude -Lbuild_lab -ldk_hip -Wl,-rpath,'$ORIGIN' -o build_lab/attn_lab
# optional ablation builds of the v3 GEMM K loop: ABL="1 2 4" -> build_lab/abl<n>/{libdk_hip.so,gemm_lab}
for n in ${ABL:-}; do
  mkdir -p build_lab/abl$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDK_LAB_ABLATIONS -DDK_V3_ABL=${n%%_*} ${ABLDEF:-} -c diffusionkit_amd/csrc/gemm256v3.hip -o build_lab/abl$n/gemm256v3.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_lab/abl$n/libdk_hip.so $(ls build_lab/obj/*.o | grep -v gemm256v3.o) build_lab/abl$n/gemm256v3.o
  cp build_lab/gemm_lab build_lab/abl$n/gemm_lab
  rm build_lab/abl$n/gemm256v3.o
done
