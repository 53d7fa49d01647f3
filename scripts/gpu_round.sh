#!/bin/bash
# One gpurun call = tests + smoke + microbench + bench + rocprof, every stage logged under gpurun_out/.
# usage: scripts/gpu_round.sh <tag> [stages...]   stages: tests smoke micro bench prof pmc
set -u
TAG=${1:-r01}; shift || true
STAGES=${@:-tests smoke micro bench prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider --timeout=600 -rf > $OUT/pytest.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest.log; tail -n 40 $OUT/pytest.log
fi
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -n 5 $OUT/smoke.log
fi
if has micro; then
  timeout 600 python scripts/microbench.py > $OUT/micro.log 2>&1; echo "micro exit $?" >> $OUT/micro.log; cat $OUT/micro.log
fi
if has bench; then
  timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 > $OUT/bench.log 2>&1; echo "bench exit $?" >> $OUT/bench.log; tail -n 5 $OUT/bench.log
fi
if has prof; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o flux -- python $OLDPWD/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OLDPWD/$OUT/prof.log 2>&1)
  echo "prof exit $?" >> $OUT/prof.log; tail -n 3 $OUT/prof.log
  python scripts/rocpd_summary.py $(find $OUT/prof -name "*.db" | head -1) --by-grid > $OUT/kernel_stats.md 2>&1; head -n 12 $OUT/kernel_stats.md
  rm -rf $OUT/prof  # the rocpd database (10+ MiB per run) stays on the box; the summary travels
fi
if has pmc; then
  # counters in their own runs (kernel-trace only), CSV output; FETCH_SIZE and WRITE_SIZE cannot share a pass
  for CNT in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"; do
    TAGC=$(echo $CNT | cut -d" " -f1)
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $CNT -d $OLDPWD/$OUT/pmc_$TAGC -o flux --output-format csv -- python $OLDPWD/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-configs > $OLDPWD/$OUT/pmc_$TAGC.log 2>&1)
    echo "pmc $TAGC exit $?"
    find $OUT/pmc_$TAGC -name "*kernel_trace.csv" -size +30M -delete
  done
  python scripts/pmc_traffic.py $OUT > $OUT/pmc_traffic.log 2>&1; tail -n 12 $OUT/pmc_traffic.log
  { python scripts/pmc_summary.py $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES gemm256v3; python scripts/pmc_summary.py $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES dk_attn; } > $OUT/pmc_sq.log 2>&1
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES  # raw CSVs exceed the 64 MiB pull limit
fi
ls -la $OUT
