for pat in random graph bytarget; do
  python scripts/bench_pyr_build.py 1024 5 512 "8 waves" $pat 2>&1 | grep -av "identical" | tail -8
done
python scripts/bench_pyr_build.py 4096 3 512 "plain order" graph 2>&1 | grep -a "index pattern\|ms per"
python scripts/bench_pyr_build.py 4096 3 512 "plain order" bytarget 2>&1 | grep -a "index pattern\|ms per"
