OUT=$1
python scripts/bench_gates.py 2>&1 | tee $OUT/bench_gates.txt
