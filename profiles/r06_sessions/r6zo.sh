export TMPDIR=/tmp
mkdir -p gpurun_out/r6zo
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-sensitivity --no-pmc --no-lowmem --no-check > gpurun_out/r6zo/c3.json 2> gpurun_out/r6zo/c3.err
python bench.py --config C5 --no-cpu-baseline --no-pmc --no-sensitivity --no-lowmem --steps 5 --warmup 2 > gpurun_out/r6zo/c5.json 2> gpurun_out/r6zo/c5.err
DH_CONV_NT_OUT=0 DH_BA_STRICT=2 DH_PYR_BUILD_DUAL=0 python bench.py --config C5 --no-cpu-baseline --no-pmc --no-sensitivity --no-lowmem --steps 5 --warmup 2 > gpurun_out/r6zo/c5_old.json 2> gpurun_out/r6zo/c5_old.err
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-sensitivity --no-pmc --no-lowmem --no-check > gpurun_out/r6zo/c3b.json 2> gpurun_out/r6zo/c3b.err
for f in c3 c5 c5_old c3b; do python - $f <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6zo/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'step %.2f op %.2f lookup %.3f ba %.3f'%(d['ms_per_step'], d['ms_update_operator'], d['ms_corr_lookup'], d['ms_per_global_ba']))
PY
done
