#!/bin/bash
# Same-box A/B of two builds of libdk_hip.so on the bench: scripts/ab_lib.sh <old.so> <new.so> [workload] [reps]
# (keep the old build with `cp diffusionkit_amd/libdk_hip.so build_lab/libdk_old.so` before rebuilding)
OLD=$(realpath $1); NEW=$(realpath $2); WL=${3:-flux-schnell-1024}; REPS=${4:-2}
for rep in $(seq $REPS); do
  for lib in $OLD $NEW; do
    DK_HIP_LIB=$lib python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib)', d['value'], 'img/s', d['denoise_ms_per_step'], 'ms/step', d['vae_decode_ms'], 'ms vae')"
  done
done
