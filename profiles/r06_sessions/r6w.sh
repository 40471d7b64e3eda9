mkdir -p gpurun_out/r6w
for i in 1 2 3; do
  DH_LIB_DIR=variant_pyrv1 python scripts/bench_pyr_build.py 256 7 64 "8 waves (rounds" 2>&1 | grep -v "^$" | sed 's/^/v1: /' >> gpurun_out/r6w/ab.txt
  python scripts/bench_pyr_build.py 256 7 64 "8 waves" 2>&1 | sed 's/^/v2: /' >> gpurun_out/r6w/ab.txt
done
cat gpurun_out/r6w/ab.txt

