#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "pyramid" 2>&1 | tail -3
for flow in reproj smooth random; do
  python scripts/bench_lookup.py --edges 1024 --flow $flow 2>&1 | grep lookup
done
