export TMPDIR=/tmp
mkdir -p gpurun_out/r6zb
python -m pytest tests -m gpu -x -q > gpurun_out/r6zb/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6zb/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6zb/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py --steps 10 --warmup 3 > gpurun_out/r6zb/bench.json 2> gpurun_out/r6zb/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6zb/bench.json').read().strip().splitlines()[-1])
for k in ['ms_per_step','ms_update_operator','ms_corr_lookup','ms_per_global_ba','ms_pyramid_build','ms_pyramid_first_build']: print(k, d.get(k))
print('lowmem', d['lowmem']['ms_per_step'], 'fg', d['factor_graph_update']['ms_per_step'], 'roofline', d['roofline']['frac'])
PY
