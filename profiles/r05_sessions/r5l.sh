#!/bin/bash
# round 5, session l: GRU epilogues with their operand loads requested up front -- operator parity, phase timeline, gate timings
OUT=$1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "accumulator_tile or update_operator or conv2d_nhwc_matches" 2>&1 | tail -n 3
timeout 150 python -u scripts/conv_timeline.py --edges 1024 --out $OUT/conv_timeline.json 2>&1 | grep "^conv3x3" | tee $OUT/conv_timeline.txt | cut -c60-400
timeout 200 python scripts/bench_gates.py --edges 4096 --reps 5 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_gates.txt | tail -n 10
