#!/bin/bash
# Round-3 GPU session D: re-run of the tests touched since session C.  Usage: gpurun --timeout 600 -- bash scripts/gpu_r3d.sh TAG
TAG=${1:-r03d}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
timeout 500 python -m pytest tests/test_policy_gpu.py tests/test_ref_parity.py::test_corr_index_double_volumes_vs_reference "tests/test_gpu_parity.py::test_update_forward_leaves_a_channel_last_hidden_state_untouched" tests/test_ref_callers_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; grep -E "passed|failed|FAILED|Error|assert" $O/pytest.log | tail -n 20
timeout 200 python scripts/debug_motion_filter.py > $O/motion_filter.log 2>&1; echo "mf rc=$?"; grep scale $O/motion_filter.log
