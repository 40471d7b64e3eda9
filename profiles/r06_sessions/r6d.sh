#!/bin/bash
# round 6, session d: the gates on 64-cout tiles at three workgroups per CU (option conv_gate64) against the product form, same box
OUT=$1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gates_on_64 or conv_64_cout or context_term" > $OUT/pytest_gate64.log 2>&1; echo "pytest rc=$?"; tail -n 5 $OUT/pytest_gate64.log
for v in 0 1 0 1; do
  echo "== bench_gates conv_gate64=$v"; DH_CONV_GATE64=$v timeout 300 python scripts/bench_gates.py --reps 5 2>&1 | tail -n 10
done
for v in 0 2; do
  echo "== bench_conv conv_gate64=$v"; DH_CONV_GATE64=$v DH_REPS=5 timeout 300 python scripts/bench_conv.py 4096 2>&1 | grep -E "3x3 128->128|heads0|320->256|448->128"
done
for v in 0 1 3 0; do
  DH_CONV_GATE64=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity --no-projection --no-product-class > $OUT/bench_g$v.log 2>&1
  echo "== bench conv_gate64=$v rc=$?"; grep '^{' $OUT/bench_g$v.log | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(d[k],3) for k in ('ms_per_step','ms_update_operator','ms_corr_lookup','ms_per_global_ba')}, (d.get('check') or {}).get('ok'))"
done
