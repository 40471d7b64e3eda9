#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "pyramid" 2>&1 | tail -2
for flow in reproj smooth; do
  timeout 120 python scripts/bench_lookup.py --edges 1024 --flow $flow 2>&1 | grep "lookup\|build"
done
timeout 120 python scripts/bench_lookup.py --edges 4096 --flow reproj 2>&1 | grep "lookup\|build"
OUT=gpurun_out/pmc_e; mkdir -p $OUT
i=0
while read -r pmc; do
  [ -z "$pmc" ] && continue
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pass$i -o run -- python scripts/bench_lookup.py --edges 1024 --reps 2 --flow reproj > $OUT/pass$i.log 2>&1 || echo "pass $i failed: $pmc"
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
LIST
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pass*/run_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "pyr_lookup" in r["Kernel_Name"]:
            k = r["Counter_Name"]; agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        print("%s %s %.5g" % (f.split("/")[-2], k, v / n))
PY
