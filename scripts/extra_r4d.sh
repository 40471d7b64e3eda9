#!/bin/bash
# session r4d extras: A/B of the 512-pixel-tile convolution (conv_halo3) -- kernel level with power / clock, then the whole step;
# the bench's multi-rank path with one rank under nccl; Cholesky step with the triangular grid (kernel stats)
OUT=$1
for v in 0 1; do
  timeout 200 python scripts/conv_power.py --seconds 3 --shapes zr,q,c128 --opt conv_halo3=$v --out $OUT/conv_power_halo3_$v.json 2>&1 | grep -v amdgpu.ids
done
for v in 0 1; do
  DH_CONV_HALO3=$v timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sensitivity --no-pmc --no-lowmem > $OUT/bench_halo3_$v.log 2>&1; echo "bench halo3=$v rc=$?"
  grep '^{' $OUT/bench_halo3_$v.log | tail -n 1 > $OUT/bench_halo3_$v.json
done
DH_BENCH_DIST1=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sensitivity --no-pmc --no-lowmem > $OUT/bench_dist1_nccl.log 2>&1; echo "dist1 rc=$?"
grep '^{' $OUT/bench_dist1_nccl.log | tail -n 1 > $OUT/bench_dist1_nccl.json
python - <<PY
import json
for f in ("bench_halo3_0", "bench_halo3_1", "bench_dist1_nccl"):
    try:
        d = json.load(open("$OUT/%s.json" % f))
        print(f, {k: round(d.get(k), 3) for k in ("ms_per_step", "ms_per_global_ba", "ms_corr_lookup", "ms_update_operator")}, d.get("check", {}).get("ok"), {k: d.get("dist", {}).get(k) for k in ("backend", "world_size", "packed_exchange", "ms_collectives_per_global_ba")} if "dist" in d else "")
    except Exception as e:
        print(f, "unreadable:", e); print(open("$OUT/%s.log" % f).read()[-1500:])
PY
