#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "pyramid" 2>&1 | tail -3
for flow in reproj smooth random; do
  timeout 120 python scripts/bench_lookup.py --edges 1024 --flow $flow --build-reps 2 2>&1 | grep "lookup\|build"
done
