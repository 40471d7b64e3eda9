#!/bin/bash
# round 5, session b: the 64-cout kernel form that was committed un-run (parity + A/B), the C / w assertions, the bench line
OUT=$1
echo "== halo64 parity (ablation build)"
DROID_HIP_TEST_ABLATION=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "64_cout or second_kernel_form" 2>&1 | tail -n 8
echo "== flow2 layer A/B at 4096 edges (ablation build): production kernel, then conv3x3_halo64_kernel"
DH_ABLATION_BUILD=1 DH_SHAPE=flow2 DH_REPS=7 DH_CHECK=1 timeout 200 python scripts/bench_conv.py 4096 2>&1 | tail -n 3
DH_ABLATION_BUILD=1 DH_CONV_HALO64=1 DH_SHAPE=flow2 DH_REPS=7 DH_CHECK=1 timeout 200 python scripts/bench_conv.py 4096 2>&1 | tail -n 3
echo "== C, w assertions"
timeout 300 python -m pytest tests/test_ref_parity.py -m gpu -q -x -k "reduced_camera" 2>&1 | tail -n 5
