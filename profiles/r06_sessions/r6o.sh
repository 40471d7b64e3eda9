#!/bin/bash
# round 6, session o: the stem as persistent 16-wave workgroups with two alternating wave groups (option conv_c7_pp): parity, step A/B
OUT=$1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv7x7" > $OUT/pytest_pp.log 2>&1; echo "pytest rc=$?"; tail -n 12 $OUT/pytest_pp.log
for v in 0 1 0 1; do
  DH_CONV_C7_PP=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity --no-projection --no-product-class > $OUT/bench_p$v.log 2>&1
  echo "== bench conv_c7_pp=$v rc=$?"; grep '^{' $OUT/bench_p$v.log | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(d[k],3) for k in ('ms_per_step','ms_update_operator','ms_corr_lookup','ms_per_global_ba')}, (d.get('check') or {}).get('ok'))"
done
DH_CONV_C7_PP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sensitivity --no-pmc --no-lowmem --no-projection --no-check --no-product-class > $OUT/prof.log 2>&1; echo "prof rc=$?"
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python scripts/kernel_stats_md.py $f | grep -E "conv7x7|kernel \|" ; find $OUT/prof -name '*kernel_trace.csv' -delete
