#!/usr/bin/env python
"""The flow encoder's 7x7 stem (conv7x7_c4_kernel and its round-6 variants) alone at C3 size: time per launch, for counter passes
(scripts/pmc_stem.sh).    python scripts/bench_stem.py [--edges 4096] [--reps 5]   (DH_CONV_C7_SPLIT / _PP / _W16 select the variants)"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT] + ([os.path.join(ROOT, "droid-slam_amd", os.environ["DH_LIB_DIR"])] if os.environ.get("DH_LIB_DIR") else []) + [os.path.join(ROOT, "droid-slam_amd")]
import torch
import droid_backends as db
from droid_amd.update import UpdateModule, EPI_RELU
from droid_amd.weights import deterministic_state_dict
from oracle import update as oupd

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=4096)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()


class _SD:
    def state_dict(self):
        return oupd.empty_state_dict()


upd = UpdateModule("cuda").load_state_dict(deterministic_state_dict(_SD(), seed=7))
E, h, w = a.edges, 48, 64
flow = torch.zeros(E, h, w, 8, device="cuda", dtype=torch.float16)
flow[..., :4] = (4 * torch.randn(E, h, w, 4, device="cuda")).half()
out = torch.empty(E, h, w, 128, device="cuda", dtype=torch.float16)
run = lambda: upd.params["flow0"]([flow], EPI_RELU, out=out)
run(); torch.cuda.synchronize()
ts = []
for _ in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[len(ts) // 2]
print("stem 7x7 4 -> 128, %d edges: %.3f ms  (%.2f TB/s of stores)" % (E, ms, E * h * w * 256 / ms / 1e9))
