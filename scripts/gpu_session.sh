#!/bin/bash
# One GPU-box session (gpurun -- bash scripts/gpu_session.sh TAG [phase ...]); everything lands under gpurun_out/TAG/.
# phases: tests  = pytest -m gpu (the whole suite) + smoke
#         quick  = pytest -m gpu on the files named in $QUICK_TESTS
#         bench  = default bench.py line (cpu baseline, in-run PMC pass)
#         prof   = rocprofv3 --kernel-trace --stats of a short bench run -> kernel_stats.md
#         power  = scripts/conv_power.py on the release build and on the -DDH_ABLATION build
#         extra  = bash $EXTRA_SCRIPT (one-off measurements of the session)
TAG=${1:-s}; shift
PHASES=${@:-tests bench prof}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
# a box whose device does not come up is given back at once (session r5h sat 300 s on one: no output, the box returned as wedged)
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); print('device ok:', torch.cuda.get_device_name(0), float(x.sum()))" \
  || { echo "device probe failed or timed out: giving the box back"; exit 3; }
for ph in $PHASES; do
  case $ph in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 25 $OUT/pytest.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log ;;
    quick)
      timeout 900 python -m pytest $QUICK_TESTS -m gpu -q -x --durations=10 > $OUT/pytest_quick.log 2>&1; echo "pytest quick rc=$?"; tail -n 25 $OUT/pytest_quick.log ;;
    bench)
      timeout 600 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"
      grep '^{' $OUT/bench.log | tail -n 1 > $OUT/bench.json
      python - <<PY
import json
try:
    d = json.load(open("$OUT/bench.json"))
    print({k: d.get(k) for k in ("ms_per_step", "ms_per_global_ba", "ms_corr_lookup", "ms_update_operator")})
    print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "traffic")}, d["roofline"].get("traffic_source", "")[:160])
    print("check", d.get("check", {}).get("ok"), "cpu", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores")})
    print("ref_py", {k: (d.get("cpu_baseline", {}).get("reference_python_c2") or {}).get(k) for k in ("value", "threads", "ba_s", "lookup_s", "update_s", "skipped")})
except Exception as e:
    print("bench line unreadable:", e); print(open("$OUT/bench.log").read()[-1500:])
PY
      ;;
    prof)
      timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sensitivity --no-pmc > $OUT/prof.log 2>&1; echo "prof rc=$?"
      f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
      [ -n "$f" ] && python scripts/kernel_stats_md.py $f > $OUT/kernel_stats.md && head -n 24 $OUT/kernel_stats.md
      find $OUT/prof -name '*kernel_trace.csv' -delete ;;
    power)
      timeout 300 python scripts/conv_power.py --out $OUT/conv_power_release.json > $OUT/conv_power_release.log 2>&1; echo "power(release) rc=$?"; tail -n 6 $OUT/conv_power_release.log
      timeout 500 python scripts/conv_power.py --ablation --out $OUT/conv_power_ablation.json > $OUT/conv_power_ablation.log 2>&1; echo "power(ablation) rc=$?"; tail -n 45 $OUT/conv_power_ablation.log ;;
    extra)
      timeout ${EXTRA_TIMEOUT:-600} bash $EXTRA_SCRIPT $OUT > $OUT/extra.log 2>&1; echo "extra rc=$?"; tail -n ${EXTRA_TAIL:-40} $OUT/extra.log ;;
  esac
done
