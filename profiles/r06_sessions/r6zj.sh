export TMPDIR=/tmp
mkdir -p gpurun_out/r6zj
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6zj/prof -o run -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-sensitivity --no-pmc --no-lowmem --no-check > gpurun_out/r6zj/prof.log 2>&1; echo "prof rc=$?"
T=$(find gpurun_out/r6zj/prof -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py "$T" --step 2 > gpurun_out/r6zj/timeline_headline_step.txt 2>&1
python scripts/step_timeline.py "$T" --step -2 > gpurun_out/r6zj/timeline_last_step.txt 2>&1
python - "$T" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
marks=[i for i,r in enumerate(rows) if "pyr_lookup_corr0_kernel" in r["Kernel_Name"]]
print("lookup launches:", len(marks))
for n,(a,b) in enumerate(zip(marks[:-1],marks[1:])):
    wall=(int(rows[b]["Start_Timestamp"])-int(rows[a]["Start_Timestamp"]))*1e-6
    print(n, "kernels %d wall %.2f ms"%(b-a, wall))
PY
rm -f "$T"
grep -v "chol_step\|chol_backsub" gpurun_out/r6zj/timeline_headline_step.txt | head -80
