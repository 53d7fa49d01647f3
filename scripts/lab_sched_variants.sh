#!/bin/bash
# lab (round 4): objects rebuilt under other AMDGPU scheduling strategies (-mllvm -amdgpu-sched-strategy=<s>) against the shipped
# build, same box, alternating.  build_lab/libdk_prev.so = the build before attention2.o took iterative-ilp; libdk_a4maxilp.so =
# shipped + attention4.o under max-ilp; libdk_max-ilp.so = both attention kernels and both 256^2 GEMMs under max-ilp.
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention" -p no:cacheprovider 2>&1 | tail -n 2
B="--no-cpu-baseline --no-other-configs"
for r in 1 2; do for L in build_lab/libdk_prev.so diffusionkit_amd/libdk_hip.so; do echo "== sd3 $L"; DK_HIP_LIB=$L python bench.py --workload sd3-medium-1024 --steps 2 --warmup 1 $B 2>/dev/null | python scripts/bench_line.py /dev/stdin; done; done
for r in 1 2; do for L in diffusionkit_amd/libdk_hip.so build_lab/libdk_a4maxilp.so; do echo "== fp8 $L"; DK_HIP_LIB=$L python bench.py --workload flux-dev-1024 --fp8 --steps 1 --warmup 1 $B 2>/dev/null | python scripts/bench_line.py /dev/stdin; done; done
for r in 1 2; do for L in diffusionkit_amd/libdk_hip.so build_lab/libdk_a4maxilp.so; do echo "== flux $L"; DK_HIP_LIB=$L python bench.py --steps 5 --warmup 2 $B 2>/dev/null | python scripts/bench_line.py /dev/stdin; done; done
for L in diffusionkit_amd/libdk_hip.so build_lab/libdk_max-ilp.so diffusionkit_amd/libdk_hip.so build_lab/libdk_max-ilp.so; do echo "== $L gemm fp8"; DK_HIP_LIB=$L COLD_W=4 python scripts/gemm_fp8_bench.py 2>&1 | tail -n 1; done
