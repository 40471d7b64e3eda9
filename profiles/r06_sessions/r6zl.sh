export TMPDIR=/tmp
mkdir -p gpurun_out/r6zl
for v in 1 2; do
DH_BA_STRICT=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6zl/prof$v -o run -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-sensitivity --no-pmc --no-lowmem --no-check > gpurun_out/r6zl/prof$v.log 2>&1
T=$(find gpurun_out/r6zl/prof$v -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py "$T" --step 2 > gpurun_out/r6zl/timeline_strict$v.txt 2>&1
rm -f "$T"
echo "== strict $v"; grep -A8 "^step:" gpurun_out/r6zl/timeline_strict$v.txt
done
