#!/bin/bash
# Round-3 GPU session A: scale parity tests + full GPU suite, the default bench line (with check / sensitivity), the
# global-BA iteration with both correlation schemes, C5 on one GPU, the 2-rank self-launch, kernel stats.
# Usage: gpurun --timeout 1500 -- bash scripts/gpu_r3a.sh TAG
TAG=${1:-r03a}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; tail -n 4 $O/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench_c3.log 2>&1; echo "bench rc=$? t=$(( $(date +%s) - t0 ))"; tail -n 1 $O/bench_c3.log | cut -c1-1500
timeout 200 python bench.py --lowmem --lowmem-corr alt --steps 8 --warmup 1 > $O/lowmem_alt.log 2>&1; echo "lowmem alt rc=$?"; tail -n 1 $O/lowmem_alt.log | cut -c1-900
timeout 200 python bench.py --lowmem --lowmem-corr pyramid --steps 8 --warmup 1 > $O/lowmem_pyr.log 2>&1; echo "lowmem pyramid rc=$?"; tail -n 1 $O/lowmem_pyr.log | cut -c1-900
timeout 400 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5.log 2>&1; echo "c5 rc=$? t=$(( $(date +%s) - t0 ))"; tail -n 1 $O/bench_c5.log | cut -c1-1500
DH_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo.log 2>&1; echo "2-rank rc=$?"; tail -n 1 $O/bench_2rank_gloo.log | cut -c1-700
(cd scripts/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o ta_rate ta_rate.hip 2>/dev/null; timeout 120 ./ta_rate) > $O/ta_rate.log 2>&1; echo "ta_rate rc=$?"; tail -n 30 $O/ta_rate.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-sensitivity > $O/prof.log 2>&1; echo "prof rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lowmem -o run -- python bench.py --lowmem --lowmem-corr alt --steps 8 --warmup 1 > $O/prof_lowmem.log 2>&1; echo "prof lowmem rc=$?"
for d in prof prof_lowmem; do f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -n 14 $f | cut -d, -f1-4,7 | cut -c1-150; done
echo "total t=$(( $(date +%s) - t0 ))"
