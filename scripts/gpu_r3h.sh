#!/bin/bash
TAG=${1:-r03h}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 200 python scripts/debug_wino.py > $O/debug_wino.log 2>&1; echo "rc=$?"; grep -v amdgpu $O/debug_wino.log | head -30
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "winograd" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error|assert" $O/pytest.log | tail -n 12
DH_WINO=1 DH_SHAPE=3x3 timeout 200 python scripts/bench_conv.py 1024 > $O/bench_conv_random.log 2>&1; echo "bench_conv rc=$?"; grep -v amdgpu $O/bench_conv_random.log
DH_WINO=1 DH_SHAPE=gates DH_FILL=zero timeout 200 python scripts/bench_conv.py 1024 > $O/bench_conv_zero.log 2>&1; grep -v amdgpu $O/bench_conv_zero.log
