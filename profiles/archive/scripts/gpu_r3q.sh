#!/bin/bash
# round 3, GPU call Q: in-model A/B of the attention kernels (pipelined 7 against phase-alternating 9), bf16 and fp8
cd "$(dirname "$0")/.."
O=gpurun_out/r3q; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for rep in 1 2; do for m in 7 9; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --tune attn=$m > $O/bench_attn${m}_$rep.json 2> $O/bench_attn${m}_$rep.err; echo "attn$m rep$rep rc $?" >> $O/summary.txt
done; done
for m in 7 9; do
  timeout 300 python bench.py --fp8 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --tune attn=$m > $O/bench_fp8_attn${m}.json 2> $O/bench_fp8_attn${m}.err; echo "fp8 attn$m rc $?" >> $O/summary.txt
done
cat $O/summary.txt
for f in $O/bench_attn*.json $O/bench_fp8*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], d["value"], "img/s", d["ms_per_step"], "ms; denoise/step", d["denoise_ms_per_step"], "vae", d["vae_decode_ms"], "gemm", r.get("achieved"), "attn", (r.get("attention") or {}).get("achieved"))
except Exception as e: print(sys.argv[1], "unparsed", e)
PY
done
