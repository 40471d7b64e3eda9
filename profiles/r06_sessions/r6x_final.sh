export TMPDIR=/tmp
mkdir -p gpurun_out/r6xx
python -m pytest tests -m gpu -x -q > gpurun_out/r6xx/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6xx/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6xx/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py > gpurun_out/r6xx/bench.json 2> gpurun_out/r6xx/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6xx/prof -o run -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > gpurun_out/r6xx/prof.log 2>&1; echo "prof rc=$?"
python scripts/kernel_stats_md.py $(find gpurun_out/r6xx/prof -name '*kernel_stats.csv' | head -1) > gpurun_out/r6xx/kernel_stats.md 2>/dev/null
find gpurun_out/r6xx/prof -name '*kernel_trace.csv' -delete
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6xx/bench.json').read().strip().splitlines()[-1])
for k in ['ms_per_step','ms_update_operator','ms_corr_lookup','ms_per_global_ba','ms_pyramid_build','ms_pyramid_first_build']: print(k, d.get(k))
print('lowmem', d['lowmem']['ms_per_step'], 'fg', d['factor_graph_update']['ms_per_step'], 'roofline', d['roofline']['frac'], 'check', d['check']['ok'])
PY
head -12 gpurun_out/r6xx/kernel_stats.md
