#!/bin/bash
# round 5, session h: phase timeline of the other 3x3 launches of the operator (128 -> 128, the fused heads, the context term)
OUT=$1
export TMPDIR=/tmp
timeout 150 python -u scripts/conv_timeline.py --edges 1024 --out $OUT/conv_timeline.json 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_timeline.txt
