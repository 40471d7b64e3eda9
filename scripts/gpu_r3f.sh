#!/bin/bash
# Round-3 GPU session F: the whole GPU suite + smoke at HEAD, the default bench line, C2 line, callers, kernel stats of the
# bench, PMC passes of the lookup (HBM traffic at HEAD).  Usage: gpurun --timeout 1700 -- bash scripts/gpu_r3f.sh TAG
TAG=${1:-r03f}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
timeout 700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -n 3 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_c3.log 2>&1; echo "bench rc=$? t=$(( $(date +%s) - t0 ))"; tail -n 1 $O/bench_c3.log | cut -c1-900
timeout 200 python bench.py --config C2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.log 2>&1; echo "c2 rc=$?"; tail -n 1 $O/bench_c2.log | cut -c1-700
timeout 300 python scripts/bench_callers.py > $O/callers.log 2>&1; echo "callers rc=$?"; tail -n 1 $O/callers.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-sensitivity > $O/prof.log 2>&1; echo "prof rc=$?"
PMC_OUT=$O/pmc timeout 900 bash scripts/pmc_bench_lookup.sh > $O/pmc.log 2>&1; echo "pmc rc=$? t=$(( $(date +%s) - t0 ))"; tail -n 12 $O/pmc.log
echo "total t=$(( $(date +%s) - t0 ))"
