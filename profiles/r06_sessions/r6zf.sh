export TMPDIR=/tmp
mkdir -p gpurun_out/r6zf
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pyramid" 2>&1 | tail -1
DH_LIB_DIR=variant_pyrv1 python scripts/bench_pyr_build.py 256 3 64 "8 waves (rounds" 2>&1 | grep -a "checksum"
python scripts/bench_pyr_build.py 256 3 64 "plain order" 2>&1 | grep -a "checksum\|ms per"
for i in 1 2 3; do
  for v in "" variant_convnt; do
    DH_LIB_DIR=$v python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-sensitivity --no-pmc --no-lowmem --no-check > gpurun_out/r6zf/b_${v:-plain}_$i.json 2> gpurun_out/r6zf/b_${v:-plain}_$i.err
    python - "$v" $i <<'PY'
import json,sys
v=sys.argv[1] or 'plain'
d=json.loads(open('gpurun_out/r6zf/b_%s_%s.json'%(v,sys.argv[2])).read().strip().splitlines()[-1])
print(v, sys.argv[2], 'step %.2f op %.2f lookup %.3f ba %.3f fg %.2f'%(d['ms_per_step'], d['ms_update_operator'], d['ms_corr_lookup'], d['ms_per_global_ba'], d['factor_graph_update']['ms_per_step']))
PY
  done
done
