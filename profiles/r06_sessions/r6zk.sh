export TMPDIR=/tmp
mkdir -p gpurun_out/r6zk
python -m pytest tests -m gpu -x -q -k "ba or graph or capi or c_abi or dist or policy or callers" 2>&1 | tail -2
for i in 1 2 3; do
  for v in 2 1; do
    DH_BA_STRICT=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sensitivity --no-pmc --no-lowmem --no-check > gpurun_out/r6zk/b_${v}_$i.json 2> gpurun_out/r6zk/b_${v}_$i.err
    python - "$v" $i <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6zk/b_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
print('ba_strict', sys.argv[1], sys.argv[2], 'step %.2f op %.2f lookup %.3f ba %.3f sum %.2f fg %.2f'%(d['ms_per_step'], d['ms_update_operator'], d['ms_corr_lookup'], d['ms_per_global_ba'], d['ms_update_operator']+d['ms_corr_lookup']+d['ms_per_global_ba'], d['factor_graph_update']['ms_per_step']))
PY
  done
done
