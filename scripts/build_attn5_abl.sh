#!/bin/bash
# lab: ablation builds of attention5.hip's tile loop (scripts/gen_attn5.py A5_ABL bit mask) -> build_lab/a5_<mask>/libdk_hip.so (select with DK_HIP_LIB)
#   scripts/build_attn5_abl.sh 1 2 7 8 48 63 64
set -e
cd "$(dirname "$0")/.."
CS=diffusionkit_amd/csrc
make -C $CS -j8 > /dev/null
for m in "$@"; do  # <mask> or <name>:<A5_OPT string>[:<mask>]
  OPTS=""; MASK=$m
  if [[ "$m" == *:* ]]; then IFS=: read NAME OPTS MASK <<< "$m"; MASK=${MASK:-0}; m=$NAME; fi
  mkdir -p build_lab/a5_$m
  A5_ABL=$MASK A5_OPT=$OPTS python scripts/gen_attn5.py > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -fno-honor-nans -fno-slp-vectorize -c $CS/attention5.hip -o build_lab/a5_$m/attention5.o 2> /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_lab/a5_$m/libdk_hip.so $(ls $CS/build/*.o | grep -v attention5.o) build_lab/a5_$m/attention5.o
  rm build_lab/a5_$m/attention5.o
  echo "built build_lab/a5_$m/libdk_hip.so"
done
python scripts/gen_attn5.py > /dev/null  # the shipped body back in place
