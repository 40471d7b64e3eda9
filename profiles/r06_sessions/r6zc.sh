for pad in 0 4096 0 4096; do
  echo "pad $pad"; DH_PYR_LDS_PAD=$pad python scripts/bench_pyr_build.py 256 7 64 "8 waves (rounds" 2>&1 | grep -a "ms per"
done
