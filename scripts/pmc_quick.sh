#!/bin/bash
# quick PMC: scripts/pmc_quick.sh <tag> "<counters>" <cmd...>
TAG=$1; shift; P=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc $P -d $OUT/p1 -o lab --output-format csv -- "$@" > $OUT/p1.log 2>&1
echo "exit $?"
