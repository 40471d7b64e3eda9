#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/pmc_conv2; mkdir -p $OUT
i=0
while read -r pmc; do
  [ -z "$pmc" ] && continue
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pass$i -o run -- python scripts/bench_conv.py 512 > $OUT/pass$i.log 2>&1 || echo "pass $i failed: $pmc"
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
LIST
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pass*/run_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "conv3x3_halo_kernel<1, 128>" in r["Kernel_Name"] and r.get("Grid_Size","") == "6291456":
            agg[r["Counter_Name"]][0] += 1; agg[r["Counter_Name"]][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(agg.items()):
        print("%s %s %.5g (n=%d)" % (f.split("/")[-2], k, v / n, n))
PY
