#!/bin/bash
# fused lookup in the product path: full GPU suite, smoke, default bench, unfused A/B
tag=${1:-r03n}; out=gpurun_out/$tag; mkdir -p $out
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err; echo "bench rc=$?"; cat $out/bench_c3.json | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: d[k] for k in ('value','ms_per_step','ms_corr_lookup','ms_update_operator','ms_per_global_ba')}); print(d['roofline']); print(d.get('check')); print(d.get('roofline_sensitivity'))
"
timeout 600 python bench.py --steps 20 --warmup 5 --unfused-lookup --no-cpu-baseline --no-sensitivity > $out/bench_c3_unfused.json 2> $out/bench_c3_unfused.err; echo "bench rc=$?"; cat $out/bench_c3_unfused.json | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: d[k] for k in ('value','ms_per_step','ms_corr_lookup','ms_update_operator','ms_per_global_ba')}); print(d['roofline']['frac'], d.get('check'))
"
