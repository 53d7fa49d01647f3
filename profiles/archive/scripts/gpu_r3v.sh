#!/bin/bash
# round 3, GPU call V (final tree): rocprofv3 kernel stats of the bench command, the whole GPU suite, smoke, driver-style bench
cd "$(dirname "$0")/.."
O=gpurun_out/r3x; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o h -- python $R/bench.py --steps 2 --warmup 1 --no-other-configs --no-cpu-baseline > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err ); echo "prof rc $?" >> $O/summary.txt
DB=$(find $O/prof -name "*.db" | head -1)
python scripts/rocpd_summary.py "$DB" --by-grid > $O/kernel_stats.md 2> $O/kernel_stats.err; echo "summary rc $?" >> $O/summary.txt
rm -rf $O/prof
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=25 -s > $O/t_all.log 2>&1; echo "t_all rc $? in $(( $(date +%s) - t0 )) s" >> $O/summary.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/summary.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench rc $?" >> $O/summary.txt
grep -h "passed\|failed\|error" $O/t_all.log | tail -5; grep -A 12 "slowest" $O/t_all.log | head -14; tail -2 $O/smoke.log; cat $O/summary.txt; head -12 $O/kernel_stats.md | cut -c1-200
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3x/bench20.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], "img/s", d["ms_per_step"], "ms; denoise/step", d["denoise_ms_per_step"], "vae", d["vae_decode_ms"], "gemm", r["achieved"], r["frac"], "attn", r["attention"]["achieved"], "conv", r["conv"])
for k,v in (d.get("other_configs") or {}).items(): print("   ", k, v["value"], v["ms_per_step"], v["roofline"].get("frac"))
PY
