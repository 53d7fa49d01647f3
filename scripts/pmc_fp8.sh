#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the fp8 bench (one image): scripts/pmc_fp8.sh <tag>  ->  gpurun_out/<tag>/pmc_traffic.log
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $CNT -d $OLDPWD/$OUT/pmc_$CNT -o flux8 --output-format csv -- python $OLDPWD/bench.py --fp8 --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OLDPWD/$OUT/pmc_$CNT.log 2>&1)
  echo "pmc $CNT exit $?"
  find $OUT/pmc_$CNT -name "*kernel_trace.csv" -size +30M -delete
done
python scripts/pmc_traffic.py $OUT > $OUT/pmc_traffic.log 2>&1; tail -n 8 $OUT/pmc_traffic.log
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
