#!/bin/bash
# round 5, session f: per-CU phase timeline of the gate convolution, product form (two workgroups per CU) against the 512-pixel form
# (one workgroup per CU) -- the evidence behind "every one-workgroup-per-CU shape lost" 
OUT=$1
export TMPDIR=/tmp
DROID_HIP_TEST_ABLATION=1 timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "512_pixel_tile or 64_cout or conv2d_nhwc_matches" 2>&1 | tail -n 4
timeout 400 python scripts/conv_timeline.py --edges 1024 --out $OUT/conv_timeline.json 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_timeline.txt
