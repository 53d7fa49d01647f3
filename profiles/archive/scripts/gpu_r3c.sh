#!/bin/bash
# round 3, GPU call C: pipelined halo kernel -- tests, decode times under conv_halo = 1 / 3 / 2 / 0, kernel profile, bench
cd "$(dirname "$0")/.."
O=gpurun_out/r3c; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -k "vae or conv_out_image or conv3x3_gn or halo or groupnorm" -p no:cacheprovider -s > $O/vae_tests.log 2>&1; echo "vae tests rc $?" >> $O/summary.txt
for t in 1 3 2 0; do
  TUNE=conv_halo=$t N=20 timeout 200 python scripts/vae_decode_bench.py 2>/dev/null >> $O/decode_times.log
done
(cd /tmp && TUNE=conv_halo=1 N=3 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof1 -o vae -- python $OLDPWD/scripts/vae_decode_bench.py > $OLDPWD/$O/prof1.log 2>&1)
python scripts/rocpd_summary.py $(find $O/prof1 -name "*.db" | head -1) --by-grid > $O/vae_kernel_stats_halo1.md 2>&1
rm -rf $O/prof1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt
grep -h "fullsize\]\|passed\|failed" $O/vae_tests.log | tail -8; cat $O/decode_times.log; cat $O/summary.txt
sed -n 24,42p $O/vae_kernel_stats_halo1.md
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], "img/s", d["ms_per_step"], "ms; denoise/step", d["denoise_ms_per_step"], "vae", d["vae_decode_ms"], "gemm", r["achieved"], "attn", r["attention"]["achieved"], "conv", r["conv"])
PY
