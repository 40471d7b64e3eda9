#!/bin/bash
tag=${1:-r03t}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py tests/test_graph_gpu.py -q -x -k "update or glo or operator or graph or composed" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o run -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sensitivity > $out/prof.log 2>&1; echo "prof rc=$?"
python scripts/kernel_stats_md.py $out/prof/run_kernel_stats.csv > $out/kernel_stats.md; grep -n "glo_reduce\|halo2_kernel<3\|lookup_corr0" $out/kernel_stats.md
tail -n 1 $out/prof.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: d[k] for k in ('ms_per_step','ms_update_operator')}, d.get('check'))
"
rm -f $out/prof/run_kernel_trace.csv
