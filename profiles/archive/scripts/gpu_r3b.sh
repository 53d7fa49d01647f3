#!/bin/bash
# round 3, GPU call B: VAE tests after the image-tail fix, per-kernel profile of the decoder under conv_halo = 1 / 2 / 0
cd "$(dirname "$0")/.."
O=gpurun_out/r3b; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -k "vae or conv_out_image or conv3x3_gn" -p no:cacheprovider > $O/vae_tests.log 2>&1; echo "vae tests rc $?" >> $O/summary.txt
for t in 1 2 0; do
  TUNE=conv_halo=$t N=20 timeout 200 python scripts/vae_decode_bench.py >> $O/decode_times.log 2>&1
done
for t in 1 0; do
  (cd /tmp && TUNE=conv_halo=$t N=3 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof$t -o vae -- python $OLDPWD/scripts/vae_decode_bench.py > $OLDPWD/$O/prof$t.log 2>&1)
  python scripts/rocpd_summary.py $(find $O/prof$t -name "*.db" | head -1) --by-grid > $O/vae_kernel_stats_halo$t.md 2>&1
  rm -rf $O/prof$t
done
tail -3 $O/vae_tests.log; cat $O/decode_times.log; cat $O/summary.txt
head -30 $O/vae_kernel_stats_halo1.md
