#!/bin/bash
# round 5, session d: the composed C2 / C3 tests on their calibrated tolerances (deviations recorded), the sanitizer build around the
# real kernels, the ROCTX ranges under rocprofv3, and the secondary bench lines (C2, C5 on one GPU, the one-rank nccl path)
OUT=$1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -q -k "composed_update_at_c2 or composed_update_at_c3 or 72x96" > $OUT/pytest_composed.log 2>&1
echo "composed rc=$?"; tail -n 12 $OUT/pytest_composed.log; cp gpurun_out/composed_deviation.json $OUT/ 2>/dev/null
timeout 900 bash scripts/sanitize_run.sh $OUT/san > $OUT/sanitize.log 2>&1; echo "sanitize rc=$?"; tail -n 12 $OUT/sanitize.log
timeout 300 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $OUT/roctx -o run -- python scripts/roctx_demo.py > $OUT/roctx_run.log 2>&1
echo "roctx rc=$?"; tail -n 3 $OUT/roctx_run.log; python scripts/roctx_demo.py --summarise $OUT/roctx | tee $OUT/roctx_ranges.txt
find $OUT/roctx -name '*kernel_trace.csv' -delete
timeout 300 python bench.py --config C2 --steps 20 --warmup 5 --no-lowmem --no-pmc > $OUT/bench_c2.log 2>&1; echo "C2 rc=$?"; grep '^{' $OUT/bench_c2.log | tail -n 1 > $OUT/bench_c2.json
DH_BENCH_DIST1=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity > $OUT/bench_dist1.log 2>&1; echo "dist1 rc=$?"; grep '^{' $OUT/bench_dist1.log | tail -n 1 > $OUT/bench_dist1.json
timeout 400 python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity > $OUT/bench_c5.log 2>&1; echo "C5 rc=$?"; grep '^{' $OUT/bench_c5.log | tail -n 1 > $OUT/bench_c5.json
python - <<PY
import json
for n in ("bench_c2", "bench_dist1", "bench_c5"):
    try:
        d = json.load(open("$OUT/%s.json" % n))
        print(n, {k: round(d[k], 3) for k in ("ms_per_step", "ms_update_operator", "ms_corr_lookup", "ms_per_global_ba")}, "check", (d.get("check") or {}).get("ok"), "frac", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(n, "unreadable:", e)
PY
