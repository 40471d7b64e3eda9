#!/bin/bash
# round 6, session k: GraphAgg's eta head as the fused second layer of agg.conv2 (option eta_fused): parity, step A/B; the 2-rank gloo self-test
OUT=$1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_graph_gpu.py -m gpu -q -x > $OUT/pytest_eta.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest_eta.log
for v in 1 0 1 0; do
  DH_ETA_FUSED=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity --no-projection --no-product-class > $OUT/bench_e$v.log 2>&1
  echo "== bench eta_fused=$v rc=$?"; grep '^{' $OUT/bench_e$v.log | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(d[k],3) for k in ('ms_per_step','ms_update_operator','ms_corr_lookup','ms_per_global_ba')}, (d.get('check') or {}).get('ok'))"
done
DH_BENCH_BACKEND=gloo timeout 400 python bench.py --config C2 --gpus 2 --steps 5 --warmup 2 > $OUT/bench_gloo2.log 2>&1; echo "gloo2 rc=$?"; grep '^{' $OUT/bench_gloo2.log | tail -n 1 | cut -c1-400
