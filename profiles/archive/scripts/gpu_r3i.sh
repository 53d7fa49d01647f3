#!/bin/bash
# round 3, GPU call I: D = 512 flash attention of the VAE mid block -- tests, decode A/B
cd "$(dirname "$0")/.."
O=gpurun_out/r3i; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -m gpu -k "d512 or vae" -p no:cacheprovider -s > $O/tests.log 2>&1; echo "tests rc $?" >> $O/summary.txt
for t in "vae_attn=1" "vae_attn=0"; do TUNE=$t N=20 timeout 200 python scripts/vae_decode_bench.py 2>/dev/null >> $O/decode_times.log; done
(cd /tmp && N=3 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o vae -- python $OLDPWD/scripts/vae_decode_bench.py > $OLDPWD/$O/prof.log 2>&1)
python scripts/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) --by-grid > $O/vae_kernel_stats.md 2>&1; rm -rf $O/prof
grep -h "fullsize\]\|passed\|failed\|Error" $O/tests.log | tail -8; cat $O/decode_times.log $O/summary.txt; grep "attn512\|softmax\|transpose" $O/vae_kernel_stats.md | head
