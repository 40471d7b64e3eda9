#!/bin/bash
# round 5, session g: the context term in the accumulator-tile layout -- parity (bit-identical operator), the operator tests, the
# phase timeline and the gate A/B with both forms, then the bench line
OUT=$1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "accumulator_tile or update_operator or conv2d_nhwc_matches" 2>&1 | tail -n 6
DROID_HIP_TEST_ABLATION=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "accumulator_tile or 512_pixel_tile" 2>&1 | tail -n 3
timeout 300 python scripts/conv_timeline.py --edges 1024 --out $OUT/conv_timeline.json 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_timeline.txt
timeout 300 python scripts/bench_gates.py --edges 4096 --reps 5 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_gates.txt
