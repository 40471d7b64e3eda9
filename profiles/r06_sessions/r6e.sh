#!/bin/bash
# round 6, session e: the next iteration's global-context reduction inside the q gate (dh_conv2d_nhwc_f16_ex3): parity, step A/B; the
# phase timeline of the three-workgroup gate form (conv_gate64) against the product form
OUT=$1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_graph_gpu.py -m gpu -q -x > $OUT/pytest_glo.log 2>&1; echo "pytest rc=$?"; tail -n 8 $OUT/pytest_glo.log
for v in chain nochain chain nochain; do
  fl=""; [ $v = nochain ] && fl="--no-glo-chain"
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity --no-projection $fl > $OUT/bench_$v.log 2>&1
  echo "== bench $v rc=$?"; grep '^{' $OUT/bench_$v.log | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(d[k],3) for k in ('ms_per_step','ms_update_operator','ms_corr_lookup','ms_per_global_ba')}, (d.get('check') or {}), {k: round(v,3) for k,v in (d.get('factor_graph_update') or {}).items() if isinstance(v,float)})"
done
timeout 400 python scripts/conv_timeline.py --gate64 --edges 1024 --out $OUT/conv_timeline_gate64.json > $OUT/conv_timeline_gate64.txt 2>&1; echo "timeline rc=$?"; cut -c1-400 $OUT/conv_timeline_gate64.txt | tail -n 30
