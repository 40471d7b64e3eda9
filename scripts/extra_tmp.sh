#!/bin/bash
OUT=$1
for v in 1 0 1; do
  DH_LOOKUP_MIX=$v timeout 200 python scripts/bench_lookup.py --edges 4096 --reps 7 --flow reproj --fused 2>&1 | grep -i "fused lookup+corr0\|again\|variant 6" | sed "s/^/mix=$v  /"
done
