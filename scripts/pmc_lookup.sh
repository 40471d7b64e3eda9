#!/bin/bash
# PMC passes over the lookup micro-benchmark (separate passes; --kernel-trace only)
export TMPDIR=/tmp
TAG=${1:-p}; EDGES=${2:-1024}
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
python scripts/bench_lookup.py --edges $EDGES --flow reproj 2>&1 | grep lookup
i=0
while read -r pmc; do
  [ -z "$pmc" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pass$i -o run -- python scripts/bench_lookup.py --edges $EDGES --reps 2 --flow reproj > $OUT/pass$i.log 2>&1 || echo "pass $i failed: $pmc"
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_CYCLE_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum
GRBM_GUI_ACTIVE
LIST
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pass*/run_counter_collection.csv"), key=lambda p: int(p.split("pass")[1].split("/")[0])):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "pyr_lookup" in r["Kernel_Name"]:
            k = r["Counter_Name"]; agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        print("%s %s %.5g" % (f.split("/")[-2], k, v / n))
PY
