#!/bin/bash
# session r4b extras: power / clock evidence of the gate convolution (rocm-smi sampler + GRBM_GUI_ACTIVE pass) and the 2-rank self-test
OUT=$1
timeout 200 python scripts/conv_power.py --ablation --seconds 4 --masks 0,15,16 --shapes zr --out $OUT/conv_power_zr.json 2>&1 | grep -v amdgpu.ids; echo "power zr rc=$?"
timeout 100 python scripts/conv_power.py --ablation --seconds 4 --masks 0 --shapes c128,q --fills randn --out $OUT/conv_power_other.json 2>&1 | grep -v amdgpu.ids
timeout 500 bash scripts/pmc_conv_power.sh $OUT/pmc_conv_power; echo "pmc rc=$?"
DH_BENCH_BACKEND=gloo timeout 500 python bench.py --gpus 2 --steps 8 --warmup 3 --no-cpu-baseline --no-sensitivity --no-pmc > $OUT/bench_2rank_gloo.log 2>&1; echo "2-rank rc=$?"
grep '^{' $OUT/bench_2rank_gloo.log | tail -n 1 > $OUT/bench_2rank_gloo.json
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_2rank_gloo.json"))
    print({k: d.get(k) for k in ("ms_per_step", "ms_per_global_ba", "ms_corr_lookup", "ms_update_operator")}, {k: d.get("dist", {}).get(k) for k in ("allreduce_bytes_per_gn_iteration", "packed_exchange", "ms_allreduce_system", "ms_allreduce_disps", "ms_collectives_per_global_ba")}, d.get("check", {}).get("ok"))
except Exception as e:
    print("unreadable:", e); print(open("$OUT/bench_2rank_gloo.log").read()[-2000:])
PY
