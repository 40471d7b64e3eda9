#!/bin/bash
# round 6, session j: the update operator on overlapping 64-column strips for images above 64 x 64 (UpdateModule._forward_strips)
OUT=$1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -q -x -k "canvas_pyramid_and_operator or big or strips or 72" --durations=8 > $OUT/pytest_strips.log 2>&1; echo "pytest rc=$?"; tail -n 16 $OUT/pytest_strips.log
python - <<'PY'
import sys, os, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "droid-slam_amd")]
import torch
from droid_amd.update import UpdateModule
from droid_amd.weights import deterministic_state_dict
from oracle import update as oupd
class _SD:
    def state_dict(self): return oupd.empty_state_dict()
upd = UpdateModule("cuda").load_state_dict(deterministic_state_dict(_SD(), seed=7))
gen = UpdateModule("cuda", canvas=False); gen.params, gen.cmap = upd.params, upd.cmap
E, K, h, w = 256, 32, 72, 96
net = torch.tanh(torch.randn(E, h, w, 128, device="cuda")).half()
inp_frames = torch.relu(torch.randn(K, h, w, 128, device="cuda")).half()
ii = (torch.arange(E, device="cuda") // 8).contiguous()
flow = torch.zeros(E, h, w, 8, device="cuda", dtype=torch.float16)
c0 = torch.relu(torch.randn(E, h, w, 128, device="cuda")).half()
for name, m in (("strips (production kernels)", upd), ("generic loop", gen)):
    for _ in range(2): m.forward_nhwc(net.clone(), None, None, flow, ii, inp_frames=inp_frames, inp_index=ii, corr0=c0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m.forward_nhwc(net, None, None, flow, ii, inp_frames=inp_frames, inp_index=ii, corr0=c0)
    torch.cuda.synchronize(); print("update operator, %d edges at %dx%d, %s: %.2f ms" % (E, h, w, name, (time.perf_counter() - t0) / 5 * 1e3))
PY
