#!/bin/bash
# SQ / LDS / TCP / TCC counters of the pyramid build kernel in separate rocprofv3 --pmc passes (kernel-trace only).
# usage: scripts/pmc_pyr_build.sh [outdir] [frames=512] [variant substring]
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_pyr}; F=${2:-512}; V=${3:-"row-pair-major, 8 waves (rounds"}; mkdir -p $OUT
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/p$i -o run -- python scripts/bench_pyr_build.py 256 1 $F "$V" > $OUT/p$i.log 2>&1 || echo "pass $i failed: $pmc"
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob("$OUT/p*/run_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "pyr_build_ring" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    for r in rows:
        if int(r["Dispatch_Id"]) != ids[0]:          # the first launch is the script's one-edge shape probe
            k = r["Counter_Name"]; agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
res = {k: v / n for k, (n, v) in agg.items()}
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
PY
find $OUT -name '*kernel_trace.csv' -delete
