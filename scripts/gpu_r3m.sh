#!/bin/bash
# fused lookup + corr0: parity test, timing
tag=${1:-r03m}; out=gpurun_out/$tag; mkdir -p $out
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "lookup_fused" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log
timeout 300 python scripts/bench_lookup.py --edges 4096 --reps 7 --fused > $out/fused.log 2>&1; echo "bench rc=$?"; tail -5 $out/fused.log
