#!/bin/bash
# Resolution / batch sweep of the bench legs (the dispatch rules were fitted on 1024 x 1024, batch 1): one line per point.
#   scripts/res_sweep.sh <outdir>
OUT=${1:-gpurun_out/sweep}; mkdir -p $OUT
run() {  # name, bench arguments
  local name=$1; shift
  timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-other-configs "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - "$name" $OUT/$name.json <<'PY'
import json, sys
name, path = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f"| {name} | {d['value']:.3f} | {d['ms_per_step']:.1f} | {d['denoise_ms_per_step']:.2f} | {d['vae_decode_ms']:.2f} | {r['achieved']:.0f} ({r['frac']:.3f}) | {r['attention']['achieved']:.0f} | {r['conv']['achieved']:.0f} | {d['mfma_roofline_frac_whole_path']:.3f} |", flush=True)
except Exception as e:  # noqa: BLE001
    print(f"| {name} | failed: {e} |", flush=True)
PY
}
echo "| leg | images/s | ms / image | denoise ms / step | decode ms | block-Linear GEMM TF (of 2500) | attention TF | conv TF | whole path |"
echo "|---|---|---|---|---|---|---|---|---|"
for R in 512 768 1024 1280; do run flux_$R --workload flux-schnell-1024 --res $R --steps 4 --warmup 2; done
for B in 2 4 8; do run flux_512_b$B --workload flux-schnell-512 --batch $B --steps 3 --warmup 1; done
for B in 2 8; do run flux_1024_b$B --workload flux-schnell-1024 --batch $B --steps 2 --warmup 1; done
for R in 512 768 1024; do run sd3_$R --workload sd3-medium-1024 --res $R --steps 2 --warmup 1; done
run sd35_512 --workload sd35-large-1024 --res 512 --steps 1 --warmup 1
