#!/bin/bash
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -x -q -k "ba or dist or capi or raw" 2>&1 | tail -3
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ba -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-update-op --no-lookup > gpurun_out/prof_ba.log 2>&1
grep -o '"ms_per_global_ba": [0-9.]*' gpurun_out/prof_ba.log
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/prof_ba/run_kernel_stats.csv')))[:8]:
    print("%-60s %5s %9.3f ms avg %8.1f us" % (r['Name'].replace('(anonymous namespace)::','').split('(')[0][:60], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
