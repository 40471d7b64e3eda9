#!/bin/bash
# round 5, session e: composed tests on the measured floors (both iterations recorded), ROCTX ranges with the rocprofiler-sdk roctx
# library, the sanitizer around the real kernels (with diagnostics), the C2 line
OUT=$1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -q -k "composed_update_at_c2 or composed_update_at_c3 or 72x96" > $OUT/pytest_composed.log 2>&1
echo "composed rc=$?"; tail -n 6 $OUT/pytest_composed.log; cp gpurun_out/composed_deviation.json $OUT/ 2>/dev/null
timeout 300 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $OUT/roctx -o run -- python scripts/roctx_demo.py > $OUT/roctx_run.log 2>&1
echo "roctx rc=$?"; ls $OUT/roctx; python scripts/roctx_demo.py --summarise $OUT/roctx | tee $OUT/roctx_ranges.txt
find $OUT/roctx -name '*kernel_trace.csv' -delete
timeout 900 bash scripts/sanitize_run.sh $OUT/san > $OUT/sanitize.log 2>&1; echo "sanitize rc=$?"; tail -n 25 $OUT/sanitize.log
timeout 300 python bench.py --config C2 --steps 20 --warmup 5 --no-lowmem --no-pmc > $OUT/bench_c2.log 2>&1; echo "C2 rc=$?"; grep '^{' $OUT/bench_c2.log | tail -n 1 > $OUT/bench_c2.json
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_c2.json"))
    print("C2", {k: round(d[k], 3) for k in ("ms_per_step", "ms_update_operator", "ms_corr_lookup", "ms_per_global_ba")}, "check", (d.get("check") or {}).get("ok"), "cpu ref", d["cpu_baseline"].get("reference_value"))
except Exception as e:
    print("C2 unreadable:", e)
PY
