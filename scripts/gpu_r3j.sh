#!/bin/bash
TAG=${1:-r03j}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
timeout 800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 12
