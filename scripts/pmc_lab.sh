#!/bin/bash
# PMC passes over one lab invocation: scripts/pmc_lab.sh <tag> <lab args...>
# (counters in their own runs, --kernel-trace only: gpurun refuses --pmc combined with other trace domains)
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL"
P3="GRBM_GUI_ACTIVE GRBM_COUNT"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
P6="TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P -d $OUT/p$i -o lab --output-format csv -- "$@" > $OUT/p$i.log 2>&1
  echo "pass $i ($P): exit $?"
done
find $OUT -name "*.csv" | head -20
