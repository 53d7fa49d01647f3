#!/bin/bash
# round 3, GPU call E: halo kernel v3 (sched_group_barrier pipeline, straight-line halo store) -- tests, A/B against the unpipelined build, ablations, decode
cd "$(dirname "$0")/.."
O=gpurun_out/r3e; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -k "vae or conv_out_image or conv3x3_gn or halo" -p no:cacheprovider -s > $O/vae_tests.log 2>&1; echo "vae tests rc $?" >> $O/summary.txt
for n in 0 0_nosgb 1 63 32 16; do
  DK_HIP_LIB=$PWD/build_lab/halo$n/libdk_hip.so timeout 120 python scripts/conv_halo_bench.py 2>/dev/null >> $O/abl.log
done
for t in 1 3 0; do TUNE=conv_halo=$t N=20 timeout 200 python scripts/vae_decode_bench.py 2>/dev/null >> $O/decode_times.log; done
grep -h "fullsize\]\|passed\|failed" $O/vae_tests.log | tail -6; cat $O/abl.log $O/decode_times.log $O/summary.txt
