#!/bin/bash
# round 6, session h: upmask head on 128-pixel tiles (option conv_k1_half) A/B; secondary lines (C2, C5 on one GPU, one-rank nccl)
OUT=$1
export TMPDIR=/tmp
for v in 0 1 0 1; do
  DH_CONV_K1_HALF=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity --no-projection --no-product-class > $OUT/bench_k$v.log 2>&1
  echo "== bench conv_k1_half=$v rc=$?"; grep '^{' $OUT/bench_k$v.log | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(d[k],3) for k in ('ms_per_step','ms_update_operator','ms_corr_lookup','ms_per_global_ba')}, (d.get('check') or {}).get('ok'))"
done
timeout 300 python bench.py --config C2 --steps 20 --warmup 5 --no-lowmem --no-pmc > $OUT/bench_c2.log 2>&1; echo "C2 rc=$?"; grep '^{' $OUT/bench_c2.log | tail -n 1 > $OUT/bench_c2.json
DH_BENCH_DIST1=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity > $OUT/bench_dist1.log 2>&1; echo "dist1 rc=$?"; grep '^{' $OUT/bench_dist1.log | tail -n 1 > $OUT/bench_dist1.json
timeout 400 python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity --no-projection > $OUT/bench_c5.log 2>&1; echo "C5 rc=$?"; grep '^{' $OUT/bench_c5.log | tail -n 1 > $OUT/bench_c5.json
python - <<PY
import json
for n in ("bench_c2", "bench_dist1", "bench_c5"):
    try:
        d = json.load(open("$OUT/%s.json" % n))
        print(n, {k: round(d[k], 3) for k in ("ms_per_step", "ms_update_operator", "ms_corr_lookup", "ms_per_global_ba")}, "check", (d.get("check") or {}).get("ok"), "frac", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(n, "unreadable:", e)
PY
