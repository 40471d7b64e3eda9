#!/bin/bash
# round 5, session j: the step with the context term pixel-major (DH_CINIT_TILED=0) and in the accumulator-tile layout (1), alternating
# on one box
OUT=$1
export TMPDIR=/tmp
for rep in 1 2; do
  for v in 0 1; do
    DH_CINIT_TILED=$v timeout 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity --no-projection > $OUT/b_$v_$rep.log 2>&1
    python - <<PY
import json
l = [x for x in open("$OUT/b_$v_$rep.log") if x.startswith("{")]
d = json.loads(l[-1])
print("DH_CINIT_TILED=$v", {k: round(d[k], 3) for k in ("ms_per_step", "ms_update_operator", "ms_corr_lookup", "ms_per_global_ba")}, "check", d["check"]["ok"], "steady(cached ctx)", round(d["steady_state_cached_context"]["ms_per_step"], 3))
PY
  done
done | tee $OUT/cinit_tiled_step_ab.txt
