#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/pmc_conv; mkdir -p $OUT
i=0
while read -r pmc; do
  [ -z "$pmc" ] && continue
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pass$i -o run -- python scripts/bench_conv.py 512 > $OUT/pass$i.log 2>&1 || echo "pass $i failed: $pmc"
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM
TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
LIST
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pass*/run_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "conv_igemm_kernel<64, 64, 128, 1>" in r["Kernel_Name"] and int(r["Grid_Size"] if "Grid_Size" in r else 0) >= 0:
            k = (r["Counter_Name"], r.get("Grid_Size", "")); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for (k, g), (n, v) in sorted(agg.items()):
        print("%s %s grid=%s %.5g (n=%d)" % (f.split("/")[-2], k, g, v / n, n))
PY
