#!/bin/bash
OUT=$1
DROID_HIP_TEST_ABLATION=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bit_identical or winograd or conv2d_nhwc" 2>&1 | tail -5
for v in 0 1 0 1; do
  timeout 200 python scripts/conv_power.py --ablation --masks 0 --seconds 3 --shapes zr,q,c128 --fills randn --opt conv_halo4=$v --out $OUT/conv_power_halo4_$v.json 2>&1 | grep -v amdgpu.ids | sed "s/^/halo4=$v  /"
done
timeout 200 python scripts/conv_power.py --ablation --masks 0 --seconds 3 --shapes zr --fills zero --opt conv_halo4=1 --out $OUT/conv_power_halo4_zero.json 2>&1 | grep -v amdgpu.ids | sed "s/^/halo4=1  /"
