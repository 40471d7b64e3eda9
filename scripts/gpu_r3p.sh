#!/bin/bash
# Round-3 closing session at HEAD (fused lookup default): other bench lines for the record
TAG=${1:-r03p}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
cd /root/repo
line() { tail -n 1 $1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: d.get(k) for k in ('value','ms_per_step','ms_corr_lookup','ms_update_operator','ms_per_global_ba','n_gpus')}, (d.get('check') or {}).get('ok'), d['roofline'].get('frac'))
"; }
timeout 300 python bench.py --config C2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.log 2>&1; echo "c2 rc=$?"; line $O/bench_c2.log
timeout 900 python bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline --no-sensitivity > $O/bench_c5.log 2>&1; echo "c5 rc=$?"; line $O/bench_c5.log
timeout 600 python bench.py --lowmem --steps 8 --warmup 8 --no-cpu-baseline > $O/lowmem.log 2>&1; echo "lowmem rc=$?"; tail -n 1 $O/lowmem.log | cut -c1-600
DH_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/gloo2.log 2>&1; echo "gloo2 rc=$?"; line $O/gloo2.log
