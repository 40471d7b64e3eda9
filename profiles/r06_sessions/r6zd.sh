export TMPDIR=/tmp
mkdir -p gpurun_out/r6zd
DH_BENCH_DIST1=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r6zd/dist1_nccl.json 2> gpurun_out/r6zd/dist1.err; echo "dist1 rc=$?"
DH_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --config C2 --gpus 2 --steps 5 --warmup 2 > gpurun_out/r6zd/c2_gloo2.json 2> gpurun_out/r6zd/gloo2.err; echo "gloo2 rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r6zd/torchrun1.json 2> gpurun_out/r6zd/torchrun1.err; echo "torchrun1 rc=$?"
for f in dist1_nccl c2_gloo2 torchrun1; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r6zd/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d['n_gpus'], d['ms_per_step'], d['value'], d['config'].get('parallelism'), d.get('check',{}).get('ok'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -3 gpurun_out/r6zd/*.err
