OUT=$1
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lookup_fused" 2>&1 | tail -3
python scripts/bench_lookup.py --ablation --fused --edges 4096 --reps 7 2>&1 | tee $OUT/lookup_fill_ab.txt
