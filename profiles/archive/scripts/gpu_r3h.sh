#!/bin/bash
# round 3, GPU call H: LayerNorm-modulate two rows per wave (A/B against the previous build), its tests, attention tests after the pruning
cd "$(dirname "$0")/.."
O=gpurun_out/r3h; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 120 python scripts/ln_bench.py 2>/dev/null >> $O/ln.log
DK_HIP_LIB=$PWD/build_lab/ln_old/libdk_hip.so timeout 120 python scripts/ln_bench.py 2>/dev/null >> $O/ln.log
timeout 120 python scripts/ln_bench.py 2>/dev/null >> $O/ln.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -q -m gpu -k "ln_modulate or attention or modulate" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc $?" >> $O/summary.txt
cat $O/ln.log; tail -3 $O/tests.log; cat $O/summary.txt
