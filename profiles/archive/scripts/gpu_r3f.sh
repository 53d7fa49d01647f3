#!/bin/bash
# round 3, GPU call F: persistent halo kernel -- tests, A/B persist on/off, skeleton ablation, decode
cd "$(dirname "$0")/.."
O=gpurun_out/r3f; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -k "vae or conv_out_image or conv3x3_gn or halo" -p no:cacheprovider -s > $O/vae_tests.log 2>&1; echo "vae tests rc $?" >> $O/summary.txt
TUNE=conv_halo_persist=1 timeout 120 python scripts/conv_halo_bench.py 2>/dev/null >> $O/abl.log
TUNE=conv_halo_persist=0 timeout 120 python scripts/conv_halo_bench.py 2>/dev/null >> $O/abl.log
for n in 63 1; do
  DK_HIP_LIB=$PWD/build_lab/halo$n/libdk_hip.so timeout 120 python scripts/conv_halo_bench.py 2>/dev/null >> $O/abl.log
done
for t in "conv_halo=1" "conv_halo=3" "conv_halo=1,conv_halo_persist=0" "conv_halo=0"; do TUNE=$t N=20 timeout 200 python scripts/vae_decode_bench.py 2>/dev/null >> $O/decode_times.log; done
grep -h "fullsize\]\|passed\|failed\|Error" $O/vae_tests.log | tail -8; cat $O/abl.log $O/decode_times.log $O/summary.txt
