export TMPDIR=/tmp
mkdir -p gpurun_out/r6zh
for i in 1 2 3; do
  DH_CONV_NT_OUT=0 python scripts/bench_stem.py --reps 9 | sed 's/^/plain: /'
  DH_CONV_NT_OUT=1 python scripts/bench_stem.py --reps 9 | sed 's/^/nt:    /'
done
python -m pytest tests -m gpu -x -q -k "conv or update or operator or stem or upmask or graph" 2>&1 | tail -2
for i in 1 2 3; do
  for v in 0 1; do
    DH_CONV_NT_OUT=$v python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-sensitivity --no-pmc --no-lowmem --no-check > gpurun_out/r6zh/b_${v}_$i.json 2> gpurun_out/r6zh/b_${v}_$i.err
    python - "$v" $i <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6zh/b_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
print('nt_out', sys.argv[1], sys.argv[2], 'step %.2f op %.2f lookup %.3f ba %.3f fg %.2f'%(d['ms_per_step'], d['ms_update_operator'], d['ms_corr_lookup'], d['ms_per_global_ba'], d['factor_graph_update']['ms_per_step']))
PY
  done
done
