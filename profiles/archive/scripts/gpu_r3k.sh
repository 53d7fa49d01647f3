#!/bin/bash
# round 3, GPU call K: issue-overlap probe; rocprofv3 kernel stats of the bench command
cd "$(dirname "$0")/.."
O=gpurun_out/r3k; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 120 build_lab/issue_probe > $O/issue_probe.log 2>&1; echo "probe rc $?" >> $O/summary.txt
R=$PWD
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o h -- python $R/bench.py --steps 2 --warmup 1 --no-other-configs --no-cpu-baseline > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err ); echo "prof rc $?" >> $O/summary.txt
DB=$(find $O/prof -name "*.db" | head -1)
python scripts/rocpd_summary.py "$DB" --by-grid > $O/kernel_stats.md 2> $O/kernel_stats.err; echo "summary rc $?" >> $O/summary.txt
rm -rf $O/prof
cat $O/issue_probe.log; cat $O/summary.txt; head -30 $O/kernel_stats.md | cut -c1-220
