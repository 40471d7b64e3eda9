#!/bin/bash
# Effective shader clock and matrix-pipe occupancy of the gate convolution per (operand fill, ablation mask):
# rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES around
# scripts/conv_power.py --quick (one pass per fill; the kernel's template arguments carry the mask).  clock = GRBM_GUI_ACTIVE / duration.
# usage: bash scripts/pmc_conv_power.sh OUTDIR
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_conv_power}; mkdir -p $OUT
for fill in zero randn; do
  timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/$fill -o run -- \
    python scripts/conv_power.py --ablation --quick 12 --shapes zr --fills $fill --masks 0,1,6,7,15,16 > $OUT/$fill.log 2>&1 || echo "pass failed: $fill"
done
python - <<PY
import csv, glob, json, collections, re
out = {"note": "per-dispatch averages; clock_GHz = GRBM_GUI_ACTIVE / kernel duration; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE); 1024 edges, gates z|r 3x3 320->256"}
for fill in ("zero", "randn"):
    cc = glob.glob("$OUT/%s/**/*counter_collection.csv" % fill, recursive=True)
    kt = glob.glob("$OUT/%s/**/*kernel_trace.csv" % fill, recursive=True)
    if not cc or not kt:
        out[fill] = "missing"; continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        m = re.search(r"conv3x3_halo2_kernel<1, true, false, (\d+)>", r["Kernel_Name"])
        if not m:
            continue
        agg[int(m.group(1))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[int(m.group(1))]["duration_s"].append(dur.get(r["Dispatch_Id"], 0.0))
    res = {}
    for mask, c in sorted(agg.items()):
        a = {k: sum(v[2:]) / max(1, len(v[2:])) for k, v in c.items()}            # (first two launches: warm-up)
        d = a.get("duration_s", 0.0)
        res[mask] = {"ms": d * 1e3, "counters": {k: v for k, v in a.items() if k != "duration_s"},
                     "clock_GHz": a.get("GRBM_GUI_ACTIVE", 0.0) / d / 1e9 if d else None,
                     "mfma_busy": a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * a["GRBM_GUI_ACTIVE"]) if a.get("GRBM_GUI_ACTIVE") else None}
    out[fill] = res
json.dump(out, open("$OUT/pmc_conv_power.json", "w"), indent=1)
print(json.dumps({f: {m: {k: v[k] for k in ("ms", "clock_GHz", "mfma_busy")} for m, v in out[f].items()} for f in ("zero", "randn") if isinstance(out[f], dict)}, indent=1))
PY
