#!/bin/bash
# One gpurun call = any of: tests, smoke, microbenchmarks, bench lines, rocprofv3 kernel tables, PMC passes -- every stage logged under
# gpurun_out/<tag>/ (only that directory travels back; what is to be judged is copied to profiles/ afterwards).
#
#   scripts/gpu_round.sh <tag> [stages...]
#
# stages (run in the order given):
#   tests            pytest -m gpu (with -s: the full-size parity lines land in pytest.log)
#   smoke            __graft_entry__.smoke()
#   micro            scripts/microbench.py
#   cpustep          scripts/cpu_flux_step.py, ALONE (round 4's first run shared the host with pytest: 250 s instead of ~170) -> cpu_flux_step.log
#   bench[:name]     python bench.py $BENCH_<name> (default line when no name)          -> bench_<name>.json
#   prof[:name]      rocprofv3 --kernel-trace --stats over bench.py $BENCH_<name> $PROF_TAIL -> kernel_stats_<name>.md
#   pmc[:name]       FETCH_SIZE / WRITE_SIZE / SQ passes (separate runs, kernel-trace only)  -> pmc_traffic_<name>.log, pmc_sq_<name>.log
#   run:<script>     any other script of scripts/ (python or shell), logged                  -> <script>.log
#   custom[:name]    eval "$CUSTOM_<name>" (a shell command line from the environment), logged -> custom_<name>.log
# named bench argument sets (override through the environment):
#   BENCH_flux="--steps 3 --warmup 1"   BENCH_sd3="--workload sd3-medium-1024 --steps 2 --warmup 1"
#   BENCH_fp8="--workload flux-dev-1024 --fp8 --steps 2 --warmup 1"   BENCH_driver="--steps 20 --warmup 5"
set -u
TAG=${1:-r04}; shift || true
STAGES=${@:-tests smoke bench prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
: "${BENCH_flux:=--steps 3 --warmup 1}"
: "${BENCH_sd3:=--workload sd3-medium-1024 --steps 2 --warmup 1}"
: "${BENCH_fp8:=--workload flux-dev-1024 --fp8 --steps 2 --warmup 1}"
: "${BENCH_sfp8:=--fp8 --steps 5 --warmup 2}"
: "${BENCH_sd35:=--workload sd35-large-1024 --steps 1 --warmup 1}"
: "${BENCH_b8:=--batch 8 --steps 3 --warmup 1}"
: "${BENCH_driver:=--steps 20 --warmup 5}"
: "${BENCH_f512:=--workload flux-schnell-512 --steps 8 --warmup 2}"
: "${BENCH_s512:=--workload sd3-medium-512 --steps 2 --warmup 1}"
: "${BENCH_b4:=--batch 4 --steps 2 --warmup 1}"
: "${PROF_TAIL:=--no-cpu-baseline --no-roofline --no-other-configs}"
PMC_SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
bench_args() { local v="BENCH_$1"; echo "${!v}"; }
# the tree this box runs: scripts/gpu_call.sh (build container) leaves the commit in .dk_build_stamp; the library's own hash is taken here
STAMP="tree: $(cat .dk_build_stamp 2>/dev/null || echo 'no .dk_build_stamp'); libdk_hip.so sha256 $(sha256sum diffusionkit_amd/libdk_hip.so 2>/dev/null | cut -c1-16); csrc sha256 $(cat diffusionkit_amd/csrc/*.hip diffusionkit_amd/csrc/*.h diffusionkit_amd/csrc/*.inc 2>/dev/null | sha256sum | cut -c1-16); run tag $TAG"
echo "$STAMP" > $OUT/stamp.txt
for ST in $STAGES; do
  KIND=${ST%%:*}; NAME=${ST#*:}; [ "$NAME" = "$ST" ] && NAME=flux
  T0=$(date +%s)
  case $KIND in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q -s --maxfail=60 -p no:cacheprovider --timeout=600 -rf --durations=15 > $OUT/pytest.log 2>&1
      echo "pytest exit $?" >> $OUT/pytest.log; grep -E "passed|failed|error" $OUT/pytest.log | tail -n 3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -n 20 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -n 3 $OUT/smoke.log ;;
    micro)
      timeout 600 python scripts/microbench.py > $OUT/micro.log 2>&1; echo "micro exit $?" >> $OUT/micro.log; cat $OUT/micro.log ;;
    cpustep)
      timeout 900 python scripts/cpu_flux_step.py > $OUT/cpu_flux_step.log 2>&1; echo "cpustep exit $?" >> $OUT/cpu_flux_step.log; cat $OUT/cpu_flux_step.log ;;
    bench)
      timeout 1200 python bench.py --gpus 1 $(bench_args $NAME) > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.err; echo "bench $NAME exit $?"
      python scripts/bench_line.py $OUT/bench_$NAME.json ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_$NAME -o run -- python $OLDPWD/bench.py --gpus 1 $(bench_args $NAME) $PROF_TAIL > $OLDPWD/$OUT/prof_$NAME.log 2>&1)
      echo "prof $NAME exit $?"
      { echo "$STAMP"; echo; echo "command: rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 $(bench_args $NAME) $PROF_TAIL"; echo;
        python scripts/rocpd_summary.py $(find $OUT/prof_$NAME -name "*.db" | head -1) --by-grid; } > $OUT/kernel_stats_$NAME.md 2>&1
      head -n 14 $OUT/kernel_stats_$NAME.md
      rm -rf $OUT/prof_$NAME ;;  # the rocpd database (10+ MiB per run) stays on the box; the summary travels
    pmc)
      # counters in their own runs (kernel-trace only), CSV output; FETCH_SIZE and WRITE_SIZE cannot share a pass
      for CNT in FETCH_SIZE WRITE_SIZE "$PMC_SQ"; do
        TAGC=$(echo $CNT | cut -d" " -f1)
        (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $CNT -d $OLDPWD/$OUT/pmc_$TAGC -o run --output-format csv -- python $OLDPWD/bench.py --gpus 1 $(bench_args $NAME | sed -E 's/--steps [0-9]+/--steps 1/; s/--warmup [0-9]+/--warmup 0/') $PROF_TAIL > $OLDPWD/$OUT/pmc_${NAME}_$TAGC.log 2>&1)
        echo "pmc $NAME $TAGC exit $?"
        find $OUT/pmc_$TAGC -name "*kernel_trace.csv" -size +30M -delete
      done
      { echo "$STAMP"; python scripts/pmc_traffic.py $OUT; } > $OUT/pmc_traffic_$NAME.log 2>&1; tail -n 14 $OUT/pmc_traffic_$NAME.log
      [ -f $OUT/pmc_gemm_traffic.json ] && mv $OUT/pmc_gemm_traffic.json $OUT/pmc_gemm_traffic_$NAME.json
      { echo "$STAMP"; python scripts/pmc_summary.py $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES gemm256; python scripts/pmc_summary.py $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES dk_attn; } > $OUT/pmc_sq_$NAME.log 2>&1
      tail -n 12 $OUT/pmc_sq_$NAME.log
      rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES ;;  # raw CSVs exceed the 64 MiB pull limit
    run)
      case $NAME in
        *.py) timeout 900 python scripts/$NAME > $OUT/${NAME%.*}.log 2>&1 ;;
        *) timeout 900 bash scripts/$NAME $OUT > $OUT/${NAME%.*}.log 2>&1 ;;
      esac
      echo "run $NAME exit $?"; tail -n 25 $OUT/${NAME%.*}.log ;;
    custom)
      V="CUSTOM_$NAME"; echo "+ ${!V}" > $OUT/custom_$NAME.log
      timeout 1200 bash -c "${!V}" >> $OUT/custom_$NAME.log 2>&1; echo "custom $NAME exit $?"; tail -n 40 $OUT/custom_$NAME.log ;;
    *) echo "unknown stage $ST" ;;
  esac
  echo "[stage $ST: $(( $(date +%s) - T0 )) s]"
done
ls -la $OUT
