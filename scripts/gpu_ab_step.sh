#!/bin/bash
export TMPDIR=/tmp
for h in 1 0; do
DH_CONV_HALO=$h timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('halo=$h', ' '.join('%s=%.2f' % (k, d[k]) for k in ['ms_per_step','ms_per_global_ba','ms_corr_lookup','ms_update_operator']))"
done
