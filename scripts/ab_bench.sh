#!/bin/bash
# A/B of a dk_tune_set knob on the FLUX bench, alternating runs: scripts/ab_bench.sh gemm_split 0 -1 [workload]
KEY=$1; A=$2; B=$3; WL=${4:-flux-schnell-1024}
for rep in 1 2; do
  for v in $A $B; do
    python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --tune $KEY=$v 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$KEY=$v', d['value'], 'img/s', d['denoise_ms_per_step'], 'ms/step', d['vae_decode_ms'], 'ms vae')"
  done
done
