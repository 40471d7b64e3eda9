#!/bin/bash
# round 5, session o (final): the multi-rank code path of bench.py on one GPU over gloo (2 ranks), after the whole suite / bench / profile phases
OUT=$1
export TMPDIR=/tmp
DH_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 8 --warmup 2 > $OUT/bench_2rank_gloo.log 2>&1; echo "2-rank gloo rc=$?"
grep '^{' $OUT/bench_2rank_gloo.log | tail -n 1 > $OUT/bench_2rank_gloo.json
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_2rank_gloo.json"))
    print("2 ranks on one GPU (gloo):", {k: round(d[k], 3) for k in ("ms_per_step", "ms_update_operator", "ms_corr_lookup", "ms_per_global_ba")}, d["dist"]["backend"], d["dist"]["world_size"], d["dist"]["allreduce_bytes_per_gn_iteration"], d["dist"]["packed_exchange"])
except Exception as e:
    print("unreadable:", e); print(open("$OUT/bench_2rank_gloo.log").read()[-1200:])
PY
