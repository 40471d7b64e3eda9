#!/bin/bash
# Round-3 GPU session B: the policy / reference-caller tests.  Usage: gpurun --timeout 900 -- bash scripts/gpu_r3b.sh TAG
TAG=${1:-r03b}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
timeout 800 python -m pytest tests/test_policy_gpu.py tests/test_ref_callers_gpu.py tests/test_graph_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -n 60 $O/pytest.log
