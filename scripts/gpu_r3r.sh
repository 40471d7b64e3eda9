#!/bin/bash
# full GPU suite + default bench line
tag=${1:-r03r}; out=gpurun_out/$tag; mkdir -p $out
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err; echo "bench rc=$?"; cat $out/bench_c3.json | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: d[k] for k in ('value','ms_per_step','ms_corr_lookup','ms_update_operator','ms_per_global_ba')}); print(d['roofline']['frac'], d.get('check'))
"
timeout 300 python scripts/bench_conv.py > $out/conv.log 2>&1; tail -12 $out/conv.log
