#!/bin/bash
# round 6, session p: the stem with sixteen 32 px x 64 cout waves per workgroup (option conv_c7_w16: 8 waves per SIMD): parity, step A/B, kernel time
OUT=$1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv7x7" > $OUT/pytest_w16.log 2>&1; echo "pytest rc=$?"; tail -n 12 $OUT/pytest_w16.log
for v in 0 1 0 1; do
  DH_CONV_C7_W16=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity --no-projection --no-product-class > $OUT/bench_w$v.log 2>&1
  echo "== bench conv_c7_w16=$v rc=$?"; grep '^{' $OUT/bench_w$v.log | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(d[k],3) for k in ('ms_per_step','ms_update_operator','ms_corr_lookup','ms_per_global_ba')}, (d.get('check') or {}).get('ok'))"
done
for v in 0 1; do
DH_CONV_C7_W16=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$v -o run -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sensitivity --no-pmc --no-lowmem --no-projection --no-check --no-product-class > $OUT/prof$v.log 2>&1; echo "prof rc=$?"
f=$(find $OUT/prof$v -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python scripts/kernel_stats_md.py $f | grep -E "conv7x7" ; find $OUT/prof$v -name '*kernel_trace.csv' -delete
done
