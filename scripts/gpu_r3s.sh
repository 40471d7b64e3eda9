#!/bin/bash
# kernel stats of the default bench at HEAD
TAG=${1:-r03s}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
cd /root/repo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check --no-sensitivity > $O/prof.log 2>&1; echo "prof rc=$?"; tail -n 1 $O/prof.log | cut -c1-200
python scripts/kernel_stats_md.py $O/prof/run_kernel_stats.csv > $O/kernel_stats.md; head -22 $O/kernel_stats.md
rm -f $O/prof/run_kernel_trace.csv
