OUT=$1
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "update or heads or lookup_fused or conv" 2>&1 | tail -4
DROID_HIP_TEST_ABLATION=1 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lookup_fused" 2>&1 | tail -2
python -m pytest tests/test_scale_gpu.py -m gpu -q -x -k "update_operator or composed" 2>&1 | tail -3
