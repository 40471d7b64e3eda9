for i in 1 2 3; do
  python scripts/bench_stem.py --reps 9 | sed 's/^/plain: /'
  DH_LIB_DIR=variant_convnt python scripts/bench_stem.py --reps 9 | sed 's/^/nt:    /'
done
python scripts/bench_gates.py 2>&1 | tail -6 | sed 's/^/plain: /'
DH_LIB_DIR=variant_convnt python scripts/bench_gates.py 2>&1 | tail -6 | sed 's/^/nt:    /'
python scripts/bench_gates.py 2>&1 | tail -6 | sed 's/^/plain: /'
DH_LIB_DIR=variant_convnt python scripts/bench_gates.py 2>&1 | tail -6 | sed 's/^/nt:    /'
