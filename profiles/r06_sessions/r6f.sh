#!/bin/bash
# round 6, session f: two pixel tiles per workgroup for the short-main-loop layers (option conv_two_tiles): parity, layer and step A/B
OUT=$1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_pixel_tiles or heads or conv" > $OUT/pytest_two.log 2>&1; echo "pytest rc=$?"; tail -n 5 $OUT/pytest_two.log
for v in 0 1 0 1; do
  echo "== bench_conv conv_two_tiles=$v"; DH_CONV_TWO_TILES=$v DH_REPS=7 timeout 300 python scripts/bench_conv.py 4096 2>&1 | grep -E "3x3 128->128|heads0"
done
echo "== bench_conv conv_two_tiles=1 maxc=448"; DH_CONV_TWO_TILES=1 DH_CONV_TWO_TILES_MAXC=448 DH_REPS=7 timeout 300 python scripts/bench_conv.py 4096 2>&1 | grep -E "3x3 128->128|heads0|320->256|448->128"
for v in 0 1 0 1; do
  DH_CONV_TWO_TILES=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lowmem --no-pmc --no-sensitivity --no-projection --no-product-class > $OUT/bench_t$v.log 2>&1
  echo "== bench conv_two_tiles=$v rc=$?"; grep '^{' $OUT/bench_t$v.log | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(d[k],3) for k in ('ms_per_step','ms_update_operator','ms_corr_lookup','ms_per_global_ba')}, (d.get('check') or {}).get('ok'))"
done
