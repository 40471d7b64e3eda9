#!/bin/bash
# SQ / TA / L2 counters of the BA kernels (global BA of the 512-keyframe graph) in separate rocprofv3 --pmc passes.
# usage: scripts/pmc_ba.sh [kernel-substring] [outdir]
export TMPDIR=/tmp
KERN=${1:-ba_gram}; OUT=${2:-gpurun_out/pmc_ba}; mkdir -p $OUT
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/p$i -o run -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-update-op --no-lookup > $OUT/p$i.log 2>&1 || echo "pass $i failed: $pmc"
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob("$OUT/p*/run_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]:
            k = r["Counter_Name"]; agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
res = {k: v / n for k, (n, v) in agg.items()}
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/summary_$KERN.json", "w"), indent=1)
PY
