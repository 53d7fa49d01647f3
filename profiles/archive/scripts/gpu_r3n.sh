#!/bin/bash
# round 3, GPU call N: lab builds of the phase-alternating attention kernel against the pipelined one (FLUX shape), same box
cd "$(dirname "$0")/.."
O=gpurun_out/r3n; mkdir -p $O; rm -f $O/attn_abl.log
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ATTN_SHAPES=1
for n in ${VARIANTS}; do
  echo "== attn$n" >> $O/attn_abl.log
  DK_HIP_LIB=$PWD/build_lab/attn$n/libdk_hip.so timeout 120 python scripts/attn_bench.py 7 9 2>&1 | grep -v amdgpu.ids >> $O/attn_abl.log
done
cat $O/attn_abl.log
