for i in 1 2 3; do
  python scripts/bench_pyr_build.py 256 7 512 "plain order" 2>&1 | grep -a "ms per\|rror\|checksum" | sed 's/^/quad swizzle (shipped): /'
  DH_LIB_DIR=variant_pyrswz python scripts/bench_pyr_build.py 256 7 512 "plain order" 2>&1 | grep -a "ms per\|rror\|checksum" | sed 's/^/dword swizzle + 4 x b32: /'
done
