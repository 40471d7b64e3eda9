#!/bin/bash
# Round-3 GPU session G: Winograd F(2,3) prototype: parity test + A/B against the direct kernel on random and zero operands.
TAG=${1:-r03g}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "winograd or conv2d_nhwc or update_operator" > $O/pytest.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; grep -E "passed|failed|FAILED|Error|assert" $O/pytest.log | tail -n 12
DH_WINO=1 DH_SHAPE=3x3 timeout 200 python scripts/bench_conv.py 1024 > $O/bench_conv_random.log 2>&1; echo "bench_conv rc=$?"; grep -v amdgpu $O/bench_conv_random.log
DH_WINO=1 DH_SHAPE=gates DH_FILL=zero timeout 200 python scripts/bench_conv.py 1024 > $O/bench_conv_zero.log 2>&1; grep -v amdgpu $O/bench_conv_zero.log
echo "total t=$(( $(date +%s) - t0 ))"
