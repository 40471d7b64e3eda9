#!/bin/bash
# session r4a extras: the edge-sharded step with 2 ranks on the one GPU (gloo self-test of the rewritten DistBA hot path)
OUT=$1
DH_BENCH_BACKEND=gloo timeout 500 python bench.py --gpus 2 --steps 8 --warmup 3 --no-cpu-baseline --no-sensitivity --no-pmc > $OUT/bench_2rank_gloo.log 2>&1; echo "2-rank rc=$?"
grep '^{' $OUT/bench_2rank_gloo.log | tail -n 1 > $OUT/bench_2rank_gloo.json
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_2rank_gloo.json"))
    print({k: d.get(k) for k in ("ms_per_step", "ms_per_global_ba", "ms_corr_lookup", "ms_update_operator")}, d.get("dist", {}).get("allreduce_bytes_per_gn_iteration"), d.get("dist", {}).get("packed_exchange"), d.get("check", {}).get("ok"))
except Exception as e:
    print("unreadable:", e); print(open("$OUT/bench_2rank_gloo.log").read()[-2000:])
PY
