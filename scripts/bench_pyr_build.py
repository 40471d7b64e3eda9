#!/usr/bin/env python
"""A/B of the wave roles of the pyramid build kernel on one MI355X (round 6): bit-equality of the records and the time of the build
kernel alone (frame-level build of E edges into an arena: dh_corr_pyramid_build_indexed), alternating between the variants.
usage: python scripts/bench_pyr_build.py [edges=256] [rounds=5] [frames=64] [only=substring of the variant names] [pattern=random|graph|bytarget]
(bit-equality and the checksum only up to 256 edges; DH_LIB_DIR=variant_x loads a variant build of the library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT] + ([os.path.join(ROOT, "droid-slam_amd", os.environ["DH_LIB_DIR"])] if os.environ.get("DH_LIB_DIR") else []) + [os.path.join(ROOT, "droid-slam_amd")]
import numpy as np
import torch
import droid_backends as db
E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
F, h, w = (int(sys.argv[3]) if len(sys.argv) > 3 else 64), 48, 64
only = sys.argv[4] if len(sys.argv) > 4 else ""
fm = torch.randn(F, 128, h, w, device="cuda").half()
prep = db.corr_pyramid_prepare_frames(fm, h, w)
g = torch.Generator(device="cuda").manual_seed(1)
pattern = sys.argv[5] if len(sys.argv) > 5 else "random"        # random frame pairs | graph: the bench's C3-like edge list, sorted by source | bytarget: the same sorted by target
if pattern == "random":
    i1 = torch.randint(0, F, (E,), device="cuda", generator=g); i2 = torch.randint(0, F, (E,), device="cuda", generator=g)
else:
    rs = np.random.RandomState(3)
    es = [(i, j) for i in range(F) for j in range(F) if i != j and abs(i - j) <= 3]
    while len(es) < 8 * F:
        i, j = rs.randint(0, F, 2)
        if abs(i - j) > 3:
            es += [(i, j), (j, i)]
    es = sorted(es[:8 * F], key=(lambda e: (e[1], e[0])) if pattern == "bytarget" else (lambda e: e))[:E]
    i1 = torch.tensor([e[0] for e in es], device="cuda"); i2 = torch.tensor([e[1] for e in es], device="cuda")
    E = len(es)
print("index pattern:", pattern, " edges:", E, " frames:", F)
out = torch.empty(E, db.corr_pyramid_build_indexed(prep, i1[:1], i2[:1], h, w, None).shape[1], dtype=torch.float16, device="cuda")
variants = [("row-pair-major, 8 waves (rounds 2-5)", 0, 8, 0, 0), ("row-pair-major, 8 waves, an edge's workgroups on one XCD", 0, 8, 1, 0),
            ("two source blocks per workgroup (16 waves), plain order", 0, 8, 0, 1), ("two source blocks per workgroup (16 waves), one XCD per edge", 0, 8, 1, 1),
            ("row-pair-major, 4 waves", 0, 4, 0, 0), ("tile-major, 8 waves (32 px per wave)", 1, 8, 0, 0),
            ("tile-major, 8 waves, an edge's workgroups on one XCD", 1, 8, 1, 0), ("tile-major, 4 waves (64 px per wave)", 1, 4, 0, 0)]
variants = [v for v in variants if only in v[0]]
ref = None
times = {v[0]: [] for v in variants}
for r in range(rounds):
    for name, tm, waves, xcd, dual in variants:
        db.set_option("pyr_build_tm", tm); db.set_option("pyr_build_waves", waves); db.set_option("pyr_build_xcd", xcd)
        try:
            db.set_option("pyr_build_dual", dual)
        except Exception:                      # (a variant build from before the option existed)
            if dual:
                continue
        db.corr_pyramid_build_indexed(prep, i1, i2, h, w, out); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); db.corr_pyramid_build_indexed(prep, i1, i2, h, w, out); b.record(); torch.cuda.synchronize()
        times[name].append(a.elapsed_time(b))
        if r == 0 and E <= 256:            # (records of 256 edges = 6.6 GB; larger runs only time)
            cur = out.clone()
            if ref is None:
                ref = cur
                v = cur.view(torch.int16).to(torch.int64).reshape(-1)
                wt = (torch.arange(v.numel(), device="cuda") % 1000003) + 1
                print("checksum of the records (compare across library builds): %d %d" % (int(v.sum()), int((v * wt).sum())))
            print("%-62s identical to the first variant: %s" % (name, torch.equal(cur, ref)))
GB = out.numel() * 2 / 1e9
for name, ts in times.items():
    if not ts:
        continue
    t = float(np.median(ts))
    print("%-62s %.3f ms per %d edges (median of %d, min %.3f)  %.2f TB/s of records written" % (name, t, E, rounds, min(ts), GB / t))
db.set_option("pyr_build_tm", 0); db.set_option("pyr_build_waves", 8)
