#!/bin/bash
# lab: K sweep (fixed tiles, growing K) of variant builds of the bf16 GEMM: what the fixed cost per round is made of
cd "$(dirname "$0")/.."
O=gpurun_out/gemm_sweep; mkdir -p $O; rm -f $O/sweep.log
export PYTHONUNBUFFERED=1 TMPDIR=/tmp COLD_W=12 SWEEP=1
for v in ${VARIANTS}; do
  DK_HIP_LIB=$PWD/build_lab/gemm_$v/libdk_hip.so timeout 200 python scripts/gemm_bf16_bench.py 2>&1 | grep -v amdgpu.ids | sed "s#$PWD/build_lab/##" >> $O/sweep.log
done
python - <<'PY'
import re
for line in open("gpurun_out/gemm_sweep/sweep.log"):
    tag = line.split()[0]
    pts = [(int(k), float(u)) for k, u in re.findall(r"K(\d+) \d+x\d+x\d+:\s+([0-9.]+) us", line)]
    a, b = pts[:5], pts[5:]
    def fit(p):
        (k0, t0), (k1, t1) = p[-2], p[-1]
        s = (t1 - t0) / (k1 - k0)
        return s * 64, t1 - s * k1
    sa, ia = fit(a); sb, ib = fit(b)
    print(f"{tag:32s} N12288 (3 rounds): {sa/3:5.2f} us/K-tile/round, fixed {ia/3:5.1f} us/round | N3072 (1 round, 80% CUs): {sb:5.2f} us/K-tile, fixed {ib:5.1f} us   raw {[u for _, u in pts]}")
PY
