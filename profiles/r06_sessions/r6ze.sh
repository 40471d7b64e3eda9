for i in 1 2 3; do
  python scripts/bench_pyr_build.py 256 7 512 "plain order" 2>&1 | grep -a "ms per\|rror" | sed 's/^/plain: /'
  DH_LIB_DIR=variant_pyrnt python scripts/bench_pyr_build.py 256 7 512 "plain order" 2>&1 | grep -a "ms per\|rror\|checksum" | sed 's/^/nt:    /'
done
