#!/bin/bash
# alternating same-box A/B of several builds of libdk_hip.so on the bench: scripts/ab_lib3.sh <workload> <lib> <lib> [...]
WL=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    DK_HIP_LIB=$(realpath $lib) python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], 'img/s', d['denoise_ms_per_step'], 'ms/step', d['vae_decode_ms'], 'ms vae')"
  done
done
