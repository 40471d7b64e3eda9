#!/usr/bin/env python
"""A/B of the two pyramid build kernels on one MI355X: bit-equality of the built volume and timing.
usage: python scripts/check_build_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import torch
import droid_backends as db
torch.manual_seed(0)
for (E, h, w) in [(3, 48, 64), (2, 16, 32), (2, 8, 16), (256, 48, 64)]:
    f1 = torch.randn(E, 128, h, w, device="cuda").half()
    f2 = torch.randn(E, 128, h, w, device="cuda").half()
    res = {}
    for mode in ("chunk", "ring"):
        db.set_option("pyr_build_chunk", int(mode == "chunk"))
        p = db.corr_pyramid_build(f1, f2); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); p = db.corr_pyramid_build(f1, f2); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        res[mode] = (p, min(ts))
    same = torch.equal(res["chunk"][0], res["ring"][0])
    nbad = (res["chunk"][0] != res["ring"][0]).sum().item()
    print("E=%d %dx%d: identical=%s (%d differing halves)  chunk %.3f ms  ring %.3f ms" % (E, h, w, same, nbad, res["chunk"][1], res["ring"][1]))
