#!/bin/bash
# Before / after of the round-6 small-launch rules inside the model: --tune gemm_split=0 leaves every launch whole (the one-wave kernel on 60 - 120-tile
# launches: round 5's choice), the default cuts them along K.  One line per run.
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d.get('roofline') or {}
print('$1:', d['value'], 'images/s,', d['denoise_ms_per_step'], 'ms/step, GEMM', r.get('achieved'), 'TF, attention', (r.get('attention') or {}).get('achieved'))"; }
for R in 512 768; do
  for T in "" "--tune gemm_split=0"; do
    python bench.py --gpus 1 --workload flux-schnell-1024 --res $R --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs $T 2>/dev/null | line "flux $R $T"
  done
done
for T in "" "--tune gemm_split=0"; do
  python bench.py --gpus 1 --workload flux-schnell-512 --batch 2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs $T 2>/dev/null | line "flux 512 batch 2 $T"
  python bench.py --gpus 1 --workload sd3-medium-512 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $T 2>/dev/null | line "sd3 512 $T"
done
python bench.py --gpus 1 --workload flux-schnell-512 --fp8 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | line "flux 512 fp8 (policy)"
python bench.py --gpus 1 --workload flux-schnell-512 --fp8 --fp8-policy speed --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | line "flux 512 fp8 (every block)"
