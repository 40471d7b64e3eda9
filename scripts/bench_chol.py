#!/usr/bin/env python
"""A/B of the Cholesky schedules inside droid_backends.ba (option chol_lookahead: 0 two launches per block column, 1 the look-ahead
step = default, 2 the dataflow schedule of the -DDH_ABLATION build, 3 its second form with LDS-DMA operands and grouped acquires --
select it with --modes 1,2,3): results compared bit
for bit with the default schedule, median time of a global BA (itrs = 2) per configuration.
    python scripts/bench_chol.py [--ablation] [--modes 1,2] [--configs C2,C3] [--reps 7]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT] + ([os.path.join(ROOT, "droid-slam_amd", "ablation")] if "--ablation" in sys.argv else []) + [os.path.join(ROOT, "droid-slam_amd")]
import numpy as np, torch
import droid_backends as db
from droid_amd import synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--ablation", action="store_true")
ap.add_argument("--modes", default="")
ap.add_argument("--configs", default="C2,C3")
ap.add_argument("--reps", type=int, default=7)
a = ap.parse_args()
modes = [int(m) for m in a.modes.split(",") if m] or ([0, 1, 2] if db.get_option("ablation_build") else [0, 1])


def run(g, t1, mode):
    db.set_option("chol_lookahead", mode)
    d = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
    args = [d(g[k]) for k in ("intrinsics", "disps_sens", "targets", "weights", "eta", "ii", "jj")]
    ts = []
    for _ in range(a.reps):
        p, q = d(g["poses"]), d(g["disps"])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); db.ba(p, q, *args, 1, t1, 2, g["lm"], g["ep"], False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return p.cpu().numpy(), q.cpu().numpy(), sorted(ts)[len(ts) // 2]


for cfg in a.configs.split(","):
    g = syn.make_graph(cfg)
    t1 = g["n_frames"]
    ref = run(g, t1, 1)
    print("%s (%d keyframes): look-ahead %.3f ms per global BA" % (cfg, t1, ref[2]), flush=True)
    for m in modes:
        if m == 1:
            continue
        got = run(g, t1, m)
        same = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
        print("   chol_lookahead %d: %.3f ms, results %s" % (m, got[2], "identical" if same else "max |dpose| %.2e" % np.abs(ref[0] - got[0]).max()), flush=True)
db.set_option("chol_lookahead", 1)
