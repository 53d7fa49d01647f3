#!/bin/bash
# round 3, GPU call M: phase-alternating attention kernel: parity, then A/B against the pipelined kernel
cd "$(dirname "$0")/.."
O=gpurun_out/r3m; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "attention_kernel_variants or spiked_key_forces" -p no:cacheprovider -x > $O/t_attn.log 2>&1; echo "t_attn rc $?" >> $O/summary.txt
tail -5 $O/t_attn.log
timeout 300 python scripts/attn_bench.py 7 9 > $O/attn_bench.log 2>&1
cat $O/attn_bench.log | grep -v amdgpu.ids
cat $O/summary.txt
