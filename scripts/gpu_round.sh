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
  find $OUT/prof -name "*stats*" | head; find $OUT/prof -name "*kernel_trace*" -size +20M -delete
fi
if has pmc; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OLDPWD/$OUT/pmc_fetch -o flux -- python $OLDPWD/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OLDPWD/$OUT/pmc_fetch.log 2>&1)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OLDPWD/$OUT/pmc_write -o flux -- python $OLDPWD/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OLDPWD/$OUT/pmc_write.log 2>&1)
  python scripts/pmc_summary.py $OUT > $OUT/pmc_summary.log 2>&1; tail -n 20 $OUT/pmc_summary.log
fi
ls -la $OUT
