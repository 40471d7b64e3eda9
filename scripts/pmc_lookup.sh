#!/bin/bash
# PMC passes over the lookup micro-benchmark (separate passes, no tracing flags besides kernel-trace)
export TMPDIR=/tmp
TAG=${1:-p}
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
python scripts/bench_lookup.py --edges 1024 --flow reproj > $OUT/bench_reproj.log 2>&1
python scripts/bench_lookup.py --edges 1024 --flow smooth > $OUT/bench_smooth.log 2>&1
python scripts/bench_lookup.py --edges 1024 --flow random --build-reps 3 > $OUT/bench_random.log 2>&1
cat $OUT/bench_*.log
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pass$i -o run -- python scripts/bench_lookup.py --edges 1024 --reps 2 --flow reproj > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUTDIR", "")
for f in sorted(glob.glob("gpurun_out/pmc_*/pass*/run_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "pyr_lookup" in r["Kernel_Name"]:
            k = r["Counter_Name"]; agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        print("%s %s per-dispatch %.4g (n=%d)" % (f.split("/")[-2], k, v / n, n))
PY
