#!/bin/bash
# HBM traffic of the lookup kernel at C3 size (4096 edges), both output variants: FETCH_SIZE / WRITE_SIZE in
# separate rocprofv3 --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2).
export TMPDIR=/tmp
OUT=gpurun_out/pmc_c3; mkdir -p $OUT
for variant in "" "--nhwc"; do
  tag=${variant:+nhwc}; tag=${tag:-nchw}
  timeout 200 python scripts/bench_lookup.py --edges 4096 --flow reproj $variant 2>&1 | grep lookup
  for pmc in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    name=$(echo $pmc | cut -d' ' -f1)
    timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/${tag}_$name -o run -- python scripts/bench_lookup.py --edges 4096 --reps 2 --flow reproj $variant > $OUT/${tag}_$name.log 2>&1 || echo "pass failed: $tag $pmc"
  done
done
python - <<PY
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob("$OUT/*/run_counter_collection.csv")):
    tag = f.split("/")[-2].split("_")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "pyr_lookup" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][0] += 1; agg[r["Counter_Name"]][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        res.setdefault(tag, {})[k] = v / n
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
PY
