#!/bin/bash
# kernel stats of the default bench (fused lookup) + PMC passes of the three lookup variants
TAG=${1:-r03o}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
cd /root/repo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check --no-sensitivity > $O/prof.log 2>&1; echo "prof rc=$?"; tail -n 1 $O/prof.log | cut -c1-300
python scripts/kernel_stats_md.py $O/prof/run_kernel_stats.csv > $O/kernel_stats.md 2>/dev/null || ls $O/prof | head
head -25 $O/kernel_stats.md
PMC_OUT=$O/pmc timeout 1200 bash scripts/pmc_bench_lookup.sh > $O/pmc.log 2>&1; echo "pmc rc=$?"; tail -n 30 $O/pmc.log
