OUT=$1
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_ref_parity.py tests/test_dist_gpu.py -m gpu -q -x -k "ba or cholesky or dist or sharded" 2>&1 | tail -3
DROID_HIP_TEST_ABLATION=1 timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cholesky_schedules" 2>&1 | tail -2
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
