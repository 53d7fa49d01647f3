#!/bin/bash
# Round-2 GPU stages; one gpurun call runs the stages named on the command line, every stage under its own timeout and logged
# under gpurun_out/<tag>/.   usage: scripts/gpu_r2.sh <tag> stage [stage ...]
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # run <name> <timeout> <cmd...>
  local name=$1 t=$2; shift 2
  local t0=$(date +%s)
  timeout $t "$@" > $OUT/$name.log 2>&1
  local rc=$?
  echo "[$name] exit $rc in $(( $(date +%s) - t0 )) s" | tee -a $OUT/$name.log
}
for st in "$@"; do
  case $st in
    probe) run fp8_probe 120 build_lab/fp8_probe; cat $OUT/fp8_probe.log ;;
    t_fp8) run t_fp8 900 python -m pytest tests/test_gpu_fp8.py -q -x -s -p no:cacheprovider --timeout=600 -rf; tail -n 30 $OUT/t_fp8.log ;;
    t_fp8_all) run t_fp8 900 python -m pytest tests/test_gpu_fp8.py -q -s -p no:cacheprovider --timeout=600 -rf; tail -n 40 $OUT/t_fp8.log ;;
    t_full) run t_full 1500 python -m pytest tests/test_gpu_fullsize.py -q -s -p no:cacheprovider --timeout=900 -rf; grep -E "fullsize|passed|failed|Error|error" $OUT/t_full.log | tail -n 30 ;;
    t_sel) run t_sel 900 python -m pytest tests/test_gpu_model.py tests/test_reference_mlx_golden.py -m gpu -q -p no:cacheprovider --timeout=600 -rf -k "denoise or sd3 or multi_seed or pipeline"; tail -n 15 $OUT/t_sel.log ;;
    t_all) run t_all 2400 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --timeout=900 -rf --durations=15; tail -n 45 $OUT/t_all.log ;;
    smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()"; tail -n 3 $OUT/smoke.log ;;
    bench) run bench 900 python bench.py --gpus 1 --steps 5 --warmup 2; tail -n 3 $OUT/bench.log ;;
    bench20) run bench20 900 python bench.py --gpus 1 --steps 20 --warmup 5; tail -n 3 $OUT/bench20.log ;;
    bench_dev) run bench_dev 900 python bench.py --workload flux-dev-1024 --steps 1 --warmup 1 --no-cpu-baseline; tail -n 3 $OUT/bench_dev.log ;;
    bench_fp8) run bench_fp8 900 python bench.py --workload flux-dev-1024 --fp8 --steps 1 --warmup 1 --no-cpu-baseline; tail -n 3 $OUT/bench_fp8.log ;;
    bench_fp8_schnell) run bench_fp8_schnell 900 python bench.py --fp8 --steps 5 --warmup 2 --no-cpu-baseline; tail -n 3 $OUT/bench_fp8_schnell.log ;;
    bench_sd3) run bench_sd3 900 python bench.py --workload sd3-medium-1024 --steps 1 --warmup 1 --no-cpu-baseline; tail -n 3 $OUT/bench_sd3.log ;;
    bench_b8) run bench_b8 900 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline; tail -n 3 $OUT/bench_b8.log ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o flux -- python $OLDPWD/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OLDPWD/$OUT/prof.log 2>&1)
      echo "prof exit $?" >> $OUT/prof.log; tail -n 3 $OUT/prof.log
      python scripts/rocpd_summary.py $(find $OUT/prof -name "*.db" | head -1) --by-grid > $OUT/kernel_stats.md 2>&1; head -n 14 $OUT/kernel_stats.md
      rm -rf $OUT/prof ;;
    prof_fp8)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof8 -o flux8 -- python $OLDPWD/bench.py --fp8 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OLDPWD/$OUT/prof_fp8.log 2>&1)
      echo "prof exit $?" >> $OUT/prof_fp8.log; tail -n 3 $OUT/prof_fp8.log
      python scripts/rocpd_summary.py $(find $OUT/prof8 -name "*.db" | head -1) --by-grid > $OUT/kernel_stats_fp8.md 2>&1; head -n 14 $OUT/kernel_stats_fp8.md
      rm -rf $OUT/prof8 ;;
    prof_sd3)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/profs -o sd3 -- python $OLDPWD/bench.py --workload sd3-medium-1024 --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OLDPWD/$OUT/prof_sd3.log 2>&1)
      echo "prof exit $?" >> $OUT/prof_sd3.log; tail -n 3 $OUT/prof_sd3.log
      python scripts/rocpd_summary.py $(find $OUT/profs -name "*.db" | head -1) --by-grid > $OUT/kernel_stats_sd3.md 2>&1; head -n 14 $OUT/kernel_stats_sd3.md
      rm -rf $OUT/profs ;;
    *) echo "unknown stage $st" ;;
  esac
done
