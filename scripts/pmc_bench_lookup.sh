#!/bin/bash
# HBM traffic of the lookup kernel at C3 size (4096 edges), both output variants: FETCH_SIZE / WRITE_SIZE in
# separate rocprofv3 --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2).
export TMPDIR=/tmp
OUT=${PMC_OUT:-gpurun_out/pmc_c3}; mkdir -p $OUT
for variant in ${PMC_VARIANTS:-"" "--nhwc" "--fused"}; do
  [ "$variant" = "none" ] && continue              # PMC_VARIANTS=none: only re-aggregate the CSVs already under $OUT
  tag=nchw; [ "$variant" = "--nhwc" ] && tag=nhwc; [ "$variant" = "--fused" ] && tag=fused
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
        if ("pyr_lookup_corr0_kernel<64, 0" in r["Kernel_Name"]) if tag == "fused" else ("pyr_lookup_kernel" in r["Kernel_Name"]):     # (not the timing ablations <64, 2 / 3 / 5>)
            agg[r["Counter_Name"]][0] += 1; agg[r["Counter_Name"]][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        res.setdefault(tag, {})[k] = v / n
EP = 4096 * 48 * 64
out = {"note": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python scripts/bench_lookup.py --edges 4096 --flow reproj [--nhwc]; "
               "per-dispatch averages of pyr_lookup_kernel (nchw / nhwc) and pyr_lookup_corr0_kernel<64, 0> (fused) on one MI355X (scripts/pmc_bench_lookup.sh). hbm_read_bytes = "
               "TCC_EA0_RDREQ_128B*128 (+64 B for the rest); FETCH_SIZE (KB) reads exactly half of it on gfx950, as MI355X_MICROARCH.md "
               "says; hbm_write_bytes = TCC_EA0_WRREQ_64B*64 = WRITE_SIZE*1024.",
       "edge_pixels": EP, "algorithmic_bytes_per_edge_pixel": 880, "algorithmic_bytes_per_edge_pixel_fused": 744}
for tag, c in res.items():
    rd = c.get("TCC_EA0_RDREQ_128B_sum", 0) * 128 + (c.get("TCC_EA0_RDREQ_sum", 0) - c.get("TCC_EA0_RDREQ_128B_sum", 0)) * 64
    wr = c.get("TCC_EA0_WRREQ_64B_sum", 0) * 64
    out[tag] = {"counters": c, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                "hbm_bytes_per_edge_pixel": (rd + wr) / EP,
                "fetch_size_x2_plus_write_size_bytes": (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024}
print(json.dumps({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "counters"}) for k, v in out.items() if k != "note"}, indent=1))
json.dump(out, open("$OUT/lookup_pmc.json", "w"), indent=1)
PY
