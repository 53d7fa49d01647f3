#!/bin/bash
# lab: variant builds of the bf16 GEMM against each other on the FLUX / SD3 shapes, cold weights (as in the model), same box
cd "$(dirname "$0")/.."
O=gpurun_out/gemm_var; mkdir -p $O; rm -f $O/gemm_var.log
export PYTHONUNBUFFERED=1 TMPDIR=/tmp COLD_W=12 CHECK=1
for rep in 1 2; do for v in ${VARIANTS}; do
  DK_HIP_LIB=$PWD/build_lab/gemm_$v/libdk_hip.so timeout 200 python scripts/gemm_bf16_bench.py 2>&1 | grep -v amdgpu.ids | sed "s#$PWD/build_lab/##" >> $O/gemm_var.log
done; done
python - <<'PY'
import re
for line in open("gpurun_out/gemm_var/gemm_var.log"):
    tag = line.split()[0]
    us = [float(x) for x in re.findall(r":\s+([0-9.]+) us", line)]
    print(f"{tag:36s}", " ".join(f"{u:7.1f}" for u in us), f"  sum {sum(us):8.1f}")
PY
