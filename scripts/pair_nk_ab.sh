#!/bin/bash
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d.get('roofline') or {}
print('$1:', d['value'], 'images/s,', d['denoise_ms_per_step'], 'ms/step, GEMM', r.get('achieved'), 'TF')"; }
for WL in "sd3-medium-512" "sd3-medium-1024 --res 768" "sd3-medium-1024" "sd35-large-1024 --res 512"; do
  for V in 32 24 16; do
    python bench.py --gpus 1 --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --tune gemm_pair_nk=$V 2>/dev/null | line "$WL gemm_pair_nk=$V"
  done
done
