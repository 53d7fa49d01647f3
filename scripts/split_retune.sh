#!/bin/bash
# after the raw-accumulator exchange (round 6): is the break-even of the whole-launch split lower, and does the producer-chain split (stream-K's 4/5 form) pay now?
FORCED=1 python scripts/gemm_small_m_bench.py 2>&1 | grep -E "^1024"
for WL in "flux-schnell-512" "sd3-medium-512" "flux-schnell-1024 --res 768"; do
  for V in 0 24 36 48; do
    python bench.py --gpus 1 --workload $WL --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline --tune gemm_split_min=$V 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$WL gemm_split_min=$V:', d['value'], 'images/s,', d['denoise_ms_per_step'], 'ms/step')"
  done
done
