mkdir -p gpurun_out/r6za; rm -f gpurun_out/r6za/ab.txt
for i in 1 2 3; do
  DH_LIB_DIR=variant_pyrv1 python scripts/bench_pyr_build.py 256 7 64 "8 waves (rounds" 2>&1 | grep -a "ms per\|rror" | sed 's/^/v1: /' >> gpurun_out/r6za/ab.txt
  python scripts/bench_pyr_build.py 256 7 64 "plain order" 2>&1 | grep -a "ms per\|rror" | sed 's/^/v2: /' >> gpurun_out/r6za/ab.txt
  python scripts/bench_pyr_build.py 256 7 64 "8 waves (rounds" 2>&1 | grep -a "ms per\|rror" | sed 's/^/v2: /' >> gpurun_out/r6za/ab.txt
  DH_LIB_DIR=variant_pyrv3 python scripts/bench_pyr_build.py 256 7 64 "plain order" 2>&1 | grep -a "ms per\|rror" | sed 's/^/v3: /' >> gpurun_out/r6za/ab.txt
  DH_LIB_DIR=variant_pyrv3 python scripts/bench_pyr_build.py 256 7 64 "8 waves (rounds" 2>&1 | grep -a "ms per\|rror" | sed 's/^/v3: /' >> gpurun_out/r6za/ab.txt
done
cat gpurun_out/r6za/ab.txt
python scripts/bench_pyr_build.py 256 5 512 "s" 2>&1 | grep -a "ms per"
