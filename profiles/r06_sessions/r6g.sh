#!/bin/bash
# round 6, session g: the 7x7 stem as 64-cout halves at four workgroups per CU (option conv_c7_split): parity, A/B, timeline of the product form
OUT=$1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv7x7" > $OUT/pytest_c7.log 2>&1; echo "pytest rc=$?"; tail -n 5 $OUT/pytest_c7.log
for v in 0 1 0 1; do
  echo "== bench_conv conv_c7_split=$v"; DH_CONV_C7_SPLIT=$v DH_REPS=9 DH_SHAPE=flow0 timeout 300 python scripts/bench_conv.py 4096 2>&1 | grep -E "flow0"
done
timeout 300 python scripts/conv_timeline.py --only stem --edges 1024 > $OUT/conv_timeline_stem.txt 2>&1; echo "timeline rc=$?"; cut -c1-420 $OUT/conv_timeline_stem.txt | tail -n 6
for v in 0 1 0 1; do
  DH_CONV_C7_SPLIT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity --no-projection --no-product-class > $OUT/bench_s$v.log 2>&1
  echo "== bench conv_c7_split=$v rc=$?"; grep '^{' $OUT/bench_s$v.log | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(d[k],3) for k in ('ms_per_step','ms_update_operator','ms_corr_lookup','ms_per_global_ba')}, (d.get('check') or {}).get('ok'))"
done
