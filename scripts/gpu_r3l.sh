#!/bin/bash
TAG=${1:-r03l}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
DH_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_8rank_gloo.log 2>&1; echo "8-rank rc=$? t=$(( $(date +%s) - t0 ))"; grep "^{" $O/bench_8rank_gloo.log | tail -n 1 | cut -c1-2500; tail -n 5 $O/bench_8rank_gloo.log | grep -v "^{" | cut -c1-300
