#!/bin/bash
# Round-3 GPU session E: alt-correlation second form (tests + A/B + the global-BA iteration), motion filter golden.
TAG=${1:-r03e}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
timeout 500 python -m pytest tests/test_policy_gpu.py tests/test_graph_gpu.py tests/test_ref_callers_gpu.py tests/test_gpu_parity.py tests/test_ref_parity.py -m gpu -q -k "altcorr or alt or lowmem or motion or policy or backend or frontend or filler or callers or proximity" > $O/pytest.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s) - t0 ))"; grep -E "passed|failed|FAILED|Error|assert" $O/pytest.log | tail -n 20
timeout 200 python scripts/bench_altcorr.py 512 mfma > $O/altcorr.log 2>&1; echo "altcorr rc=$?"; cat $O/altcorr.log | tail -5
timeout 200 python bench.py --lowmem --lowmem-corr alt --steps 8 --warmup 1 > $O/lowmem_alt.log 2>&1; echo "lowmem alt rc=$?"; tail -n 1 $O/lowmem_alt.log | cut -c1-1200
echo "total t=$(( $(date +%s) - t0 ))"
