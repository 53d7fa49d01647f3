#!/bin/bash
# round 3, GPU call Y: end-of-round bench lines of the other north-star configurations
cd "$(dirname "$0")/.."
O=gpurun_out/r3y; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 python bench.py --workload sd3-medium-1024 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_sd3.json 2> $O/bench_sd3.err; echo "sd3 rc $?" >> $O/summary.txt
timeout 600 python bench.py --workload flux-dev-1024 --fp8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dev_fp8.json 2> $O/bench_dev_fp8.err; echo "dev fp8 rc $?" >> $O/summary.txt
timeout 300 python bench.py --fp8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_schnell_fp8.json 2> $O/bench_schnell_fp8.err; echo "schnell fp8 rc $?" >> $O/summary.txt
timeout 400 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_b8.json 2> $O/bench_b8.err; echo "b8 rc $?" >> $O/summary.txt
cat $O/summary.txt
for f in sd3 dev_fp8 schnell_fp8 b8; do python - $O/bench_$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], d["value"], "img/s", d["ms_per_step"], "ms; denoise/step", d["denoise_ms_per_step"], "vae", d["vae_decode_ms"], "gemm", r.get("achieved"), r.get("frac"), "attn", (r.get("attention") or {}).get("achieved"))
PY
done
