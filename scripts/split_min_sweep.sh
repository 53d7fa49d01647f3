#!/bin/bash
# in-model A/B of the whole-launch K split's break-even threshold (dk_tune_set gemm_split_min): ms per denoising step
for WL in sd3-medium-512 flux-schnell-512; do
  for V in 0 40 64 80 1000; do
    python bench.py --gpus 1 --workload $WL --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline --tune gemm_split_min=$V 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$WL gemm_split_min=$V:', d['value'], 'images/s,', d['denoise_ms_per_step'], 'ms/step')"
  done
done
