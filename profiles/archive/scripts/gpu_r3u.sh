#!/bin/bash
# round 3, GPU call U: GEMM fixed cost per round (K sweep), overlapped decode, batch 2
cd "$(dirname "$0")/.."
O=gpurun_out/r3u; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
SWEEP=1 COLD_W=12 timeout 200 python scripts/gemm_bf16_bench.py 2>&1 | grep -v amdgpu.ids > $O/gemm_sweep.log
SWEEP=1 COLD_W=12 EPI=gelu timeout 200 python scripts/gemm_bf16_bench.py 2>&1 | grep -v amdgpu.ids >> $O/gemm_sweep.log
tr '|' '\n' < $O/gemm_sweep.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline > $O/bench_plain.json 2> $O/bench_plain.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline --overlap-decode > $O/bench_overlap.json 2> $O/bench_overlap.err
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline --batch 2 > $O/bench_b2.json 2> $O/bench_b2.err
for f in plain overlap b2; do python - $O/bench_$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split("/")[-1], d["value"], "img/s", d["ms_per_step"], "ms; denoise/step", d["denoise_ms_per_step"], "vae", d["vae_decode_ms"])
PY
done
