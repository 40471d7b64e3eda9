mkdir -p gpurun_out/r6x
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pyramid" 2>&1 | tail -3
for i in 1 2; do
  python scripts/bench_pyr_build.py 256 7 64 "s" 2>&1 | grep -a "ms per\|identical\|rror" >> gpurun_out/r6x/ab.txt
done
cat gpurun_out/r6x/ab.txt
