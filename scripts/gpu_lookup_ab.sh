#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "pyramid or update_operator" 2>&1 | tail -2
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
for k in ['value','ms_per_step','ms_per_global_ba','ms_corr_lookup','ms_update_operator']: print(k, d[k])
print(d['roofline'])"
