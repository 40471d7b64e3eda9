export TMPDIR=/tmp
mkdir -p gpurun_out/r6zn
python bench.py --config C2 --no-cpu-baseline --no-pmc > gpurun_out/r6zn/c2.json 2> gpurun_out/r6zn/c2.err; echo "c2 rc=$?"
python bench.py --config C5 --no-cpu-baseline --no-pmc --steps 5 --warmup 2 > gpurun_out/r6zn/c5.json 2> gpurun_out/r6zn/c5.err; echo "c5 rc=$?"
for f in c2 c5; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r6zn/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d['config']['workload'][:60], 'step %.2f op %.2f lookup %.3f ba %.3f'%(d['ms_per_step'], d['ms_update_operator'], d['ms_corr_lookup'], d['ms_per_global_ba']), 'lowmem', d.get('lowmem',{}).get('ms_per_step'), 'check', d.get('check',{}).get('ok'))
except Exception as e: print(sys.argv[1], 'ERR', e); print(open('gpurun_out/r6zn/%s.err'%sys.argv[1]).read()[-800:])
PY
done
