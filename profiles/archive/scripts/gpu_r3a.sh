#!/bin/bash
# round 3, GPU call A: the new parity cases, the fused VAE stage, a first bench with other_configs, conv_halo A/B
cd "$(dirname "$0")/.."
O=gpurun_out/r3a; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv3x3_gn or conv_out_image or groupnorm or conv3x3" -p no:cacheprovider > $O/ops.log 2>&1; echo "ops rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "vae or generate_image or img2img" -p no:cacheprovider > $O/model_vae.log 2>&1; echo "model_vae rc $?" >> $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dist.py -q -m gpu -s -p no:cacheprovider -k "not full_depth_pipeline and not sd3_medium_512" > $O/fullsize.log 2>&1; echo "fullsize rc $?" >> $O/summary.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --tune conv_halo=0 > $O/bench_halo0.json 2> $O/bench_halo0.err; echo "bench_halo0 rc $?" >> $O/summary.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --tune conv_halo=1 > $O/bench_halo1.json 2> $O/bench_halo1.err; echo "bench_halo1 rc $?" >> $O/summary.txt
grep -h "fullsize\]\|passed\|failed\|error" $O/*.log | tail -60
cat $O/summary.txt
for f in $O/bench.json $O/bench_halo0.json $O/bench_halo1.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print(sys.argv[1], d["value"], "img/s", d["ms_per_step"], "ms; denoise/step", d["denoise_ms_per_step"], "vae", d["vae_decode_ms"], "gemm", r.get("achieved"), "attn", (r.get("attention") or {}).get("achieved"), "conv", (r.get("conv") or {}))
    for k,v in (d.get("other_configs") or {}).items(): print("   ", k, v["value"], v["ms_per_step"], v["roofline"].get("frac"))
except Exception as e: print(sys.argv[1], "unparsed", e)
PY
done
