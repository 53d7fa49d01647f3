#!/bin/bash
# round 3, GPU call L: attention ablation builds on the FLUX shape
cd "$(dirname "$0")/.."
O=gpurun_out/r3l; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ATTN_SHAPES=1
for n in 0 1 2 4 24 32 31 6 3 7; do
  echo "== DK3_ABL=$n" >> $O/attn_abl.log
  DK_HIP_LIB=$PWD/build_lab/attn$n/libdk_hip.so timeout 120 python scripts/attn_bench.py 7 >> $O/attn_abl.log 2>&1
done
cat $O/attn_abl.log
