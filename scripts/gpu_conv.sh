#!/bin/bash
export TMPDIR=/tmp
DH_CONV_HALO=1 timeout 400 python -m pytest tests -m gpu -x -q -k "conv2d or update_operator" 2>&1 | tail -2
DH_CONV_HALO=1 timeout 200 python scripts/bench_conv.py 1024 2>&1 | grep -v amdgpu | head -5
