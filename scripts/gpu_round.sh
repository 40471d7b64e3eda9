#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace.  Usage: gpurun -- bash scripts/gpu_round.sh TAG
TAG=${1:-r}
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1; echo "prof rc=$?"
tail -n 3 gpurun_out/pytest_$TAG.log; tail -n 3 gpurun_out/smoke_$TAG.log; tail -n 2 gpurun_out/bench_$TAG.log
find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1 | xargs -r head -n 25
