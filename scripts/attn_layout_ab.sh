#!/bin/bash
# lab (round 4): LDS slot layouts of attention4.hip (-DDK4_LAYOUT=0|1|2 builds under build_lab/attn_L<n>/) on the FLUX shapes, alternating
O=${1:-gpurun_out/attn_layout}; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ATTN_SHAPES=2
for rep in 1 2; do for L in 0 1 2; do
  echo "== layout $L (run $rep)"; DK_HIP_LIB=$PWD/build_lab/attn_L$L/libdk_hip.so python scripts/attn_bench.py 9 2>&1 | grep -v amdgpu.ids
done; done
