#!/bin/bash
# Round-3 GPU session C: policy / reference-caller / fp64 / volume-build tests, MotionFilter numbers, lookup ablation modes,
# the Winograd operand-traffic micro-benchmark.  Usage: gpurun --timeout 900 -- bash scripts/gpu_r3c.sh TAG
TAG=${1:-r03c}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
timeout 600 python -m pytest tests/test_policy_gpu.py tests/test_ref_callers_gpu.py tests/test_graph_gpu.py tests/test_ref_parity.py tests/test_gpu_parity.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 12
timeout 200 python scripts/debug_motion_filter.py > $O/motion_filter.log 2>&1; echo "mf rc=$?"; grep scale $O/motion_filter.log
timeout 300 python scripts/bench_lookup.py --edges 4096 --reps 5 --flow reproj --modes 0,1,2,3,0 > $O/lookup_modes.log 2>&1; echo "modes rc=$?"; grep lookup $O/lookup_modes.log
(cd scripts/ubench && timeout 100 ./wino_lds) > $O/wino_lds.log 2>&1; echo "wino rc=$?"; cat $O/wino_lds.log
echo "total t=$(( $(date +%s) - t0 ))"
