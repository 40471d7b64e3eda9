#!/bin/bash
# round 6, session i: GRU operands requested while the accumulators are parked (DH_EPI_EARLY, compile-time) against round 5's order
# (library variant droid-slam_amd/variant_epi_late built with -DDH_EPI_EARLY=0): parity of the product build, gates and step A/B on one box
OUT=$1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "update_operator or gates or context_term or global_context or conv" > $OUT/pytest_epi.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest_epi.log
for v in early late early late; do
  d=""; [ $v = late ] && d="variant_epi_late"
  echo "== bench_gates $v"; DH_LIB_DIR=$d timeout 300 python scripts/bench_gates.py --reps 7 2>&1 | tail -n 10 | grep -E "gru"
done
for v in early late early late; do
  d=""; [ $v = late ] && d="variant_epi_late"
  DH_LIB_DIR=$d timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity --no-projection --no-product-class > $OUT/bench_$v.log 2>&1
  echo "== bench $v rc=$?"; grep '^{' $OUT/bench_$v.log | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(d[k],3) for k in ('ms_per_step','ms_update_operator','ms_corr_lookup','ms_per_global_ba')}, (d.get('check') or {}).get('ok'))"
done
