#!/bin/bash
# round 5, session c: the whole GPU suite at the commit that promoted conv3x3_halo64_kernel / added the strips, the arena, the
# coherence fallback (every failure listed: no -x), smoke, the default bench line, the kernel profile
OUT=$1
timeout 1500 python -m pytest tests -m gpu -q --durations=12 --deselect tests/test_scale_gpu.py::test_composed_update_at_c3_matches_reference_factor_graph > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 40 $OUT/pytest.log
cp gpurun_out/composed_deviation.json $OUT/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
