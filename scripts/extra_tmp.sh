OUT=$1
python -m pytest tests/test_scale_gpu.py -m gpu -q -x -k "canvas or tum or 16_9" 2>&1 | tail -15
