#!/bin/bash
TAG=${1:-r03k}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 300 python scripts/bench_lookup.py --edges 4096 --reps 7 --flow reproj --modes 0,4,2,0,4 > $O/lookup_modes.log 2>&1; echo "modes rc=$?"; grep lookup $O/lookup_modes.log
