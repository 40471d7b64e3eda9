#!/bin/bash
TAG=${1:-r03i}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -m gpu -q -k "update or conv or winograd" > $O/pytest.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; grep -E "passed|failed|FAILED|Error|assert" $O/pytest.log | tail -n 12
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sensitivity > $O/prof.log 2>&1; echo "prof rc=$?"; tail -n 1 $O/prof.log | cut -c1-600
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); grep -E "glo_reduce|conv7x7|corr0_nchw|halo2_kernel<3" $f | cut -d, -f1-4 | cut -c1-160
echo "total t=$(( $(date +%s) - t0 ))"
