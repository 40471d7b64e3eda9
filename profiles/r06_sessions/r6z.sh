mkdir -p gpurun_out/r6z; rm -f gpurun_out/r6z/ab.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pyramid" 2>&1 | tail -3
for i in 1 2; do
  DH_LIB_DIR=variant_pyrv1 python scripts/bench_pyr_build.py 256 7 64 "8 waves (rounds" 2>&1 | grep -a "ms per\|checksum\|rror" | sed 's/^/v1: /' >> gpurun_out/r6z/ab.txt
  python scripts/bench_pyr_build.py 256 7 64 "s" 2>&1 | grep -a "ms per\|checksum\|identical\|rror" | sed 's/^/v3: /' >> gpurun_out/r6z/ab.txt
done
cat gpurun_out/r6z/ab.txt
