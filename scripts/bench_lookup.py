#!/usr/bin/env python
"""Micro-benchmark of the fused pyramid lookup (and optionally the build) on one MI355X.
usage: python scripts/bench_lookup.py [--edges 1024] [--reps 5] [--flow smooth|reproj|random] [--build-reps 0]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT] + ([os.path.join(ROOT, "droid-slam_amd", "ablation")] if "--ablation" in sys.argv else []) + [os.path.join(ROOT, "droid-slam_amd")]
import numpy as np, torch
import droid_backends as db

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=1024)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--flow", default="reproj")
ap.add_argument("--build-reps", type=int, default=0)
ap.add_argument("--nhwc", action="store_true", help="channel-last variant (feeds the update operator)")
ap.add_argument("--fused", action="store_true", help="also time corr_pyramid_lookup_corr0 (lookup + first encoder layer) against lookup + corr0_nchw")
ap.add_argument("--ablation", action="store_true", help="load the -DDH_ABLATION build (droid-slam_amd/ablation/)")
ap.add_argument("--modes", default="", help="comma list of lookup_mode values to time one after the other on the same pyramid "
                "(0 product, 1 nt tap loads, 2 no output stores, 3 no tap loads -- 2 / 3 are timing ablations with wrong results)")
a = ap.parse_args()
E, h, w = a.edges, 48, 64
torch.manual_seed(0)
f = torch.randn(64, 128, h, w, device="cuda").half()
idx1 = torch.randint(0, 64, (E,), device="cuda"); idx2 = torch.randint(0, 64, (E,), device="cuda")
f1, f2 = f[idx1].contiguous(), f[idx2].contiguous()
pyr = db.corr_pyramid_build(f1, f2)
torch.cuda.synchronize()
if a.build_reps:
    t = []
    for _ in range(a.build_reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); p2 = db.corr_pyramid_build(f1[:256], f2[:256]); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
    ms = min(t); fl = 256 * 2.0 * 128 * 3072 * (3072 + 768 + 192 + 48)
    print("build 256 edges: %.3f ms  %.1f TFLOP/s  write %.1f GB/s" % (ms, fl / ms / 1e9, 256 * 25067520 / ms / 1e6))
rng = np.random.default_rng(0)
yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
if a.flow == "random":
    c = np.stack([rng.uniform(0, w, (E, h, w)), rng.uniform(0, h, (E, h, w))], -1).astype(np.float32)
elif a.flow == "smooth":
    am = rng.uniform(-6, 6, (E, 6, 1, 1)).astype(np.float32)
    c = np.stack([xx + am[:, 0] + am[:, 1] * xx / w + am[:, 2] * yy / h, yy + am[:, 3] + am[:, 4] * xx / w + am[:, 5] * yy / h], -1)
else:
    from droid_amd import synthetic as syn
    g = syn.make_graph(syn.CONFIGS["C3"])
    d = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
    ii, jj = d(g["ii"][:E]), d(g["jj"][:E])
    c, _ = db.reproject(d(g["poses"]), d(g["disps"]), d(g["intrinsics"]), ii, jj)
    c = c.cpu().numpy()
coords = torch.as_tensor(np.ascontiguousarray(c)).cuda()
look = db.corr_pyramid_lookup_nhwc if a.nhwc else db.corr_pyramid_lookup
out = look(pyr, coords); torch.cuda.synchronize()
t = []
for _ in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = look(pyr, coords); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
for mode in [int(m) for m in a.modes.split(",") if m != ""]:
    db.set_option("lookup_mode", mode)
    look(pyr, coords); torch.cuda.synchronize()
    tm = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o2 = look(pyr, coords); e1.record(); torch.cuda.synchronize(); tm.append(e0.elapsed_time(e1))
    ok = bool(torch.equal(o2, out)) if mode < 2 else None
    print("lookup_mode %d: median %.3f ms min %.3f  (880 B/ep -> %.1f GB/s)%s" % (mode, float(np.median(tm)), min(tm), 880.0 * E * h * w / float(np.median(tm)) / 1e6,
                                                                                  "" if ok is None else "  output identical to mode 0: %s" % ok))
db.set_option("lookup_mode", 0)
ms = float(np.median(t)); nbytes = 880.0 * E * h * w
print("lookup%s %s E=%d: median %.3f ms min %.3f  -> %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
    " channel-last" if a.nhwc else "", a.flow, E, ms, min(t), nbytes / ms / 1e6, nbytes / ms / 1e6 / 80.0))
if a.fused:
    sys.path.insert(0, os.path.join(ROOT, "droid-slam_amd"))
    from droid_amd.update import pack_corr0_fused
    wgt = torch.randn(128, 196, device="cuda") * 0.05
    bias = torch.randn(128, device="cuda") * 0.3
    wpk = pack_corr0_fused(wgt)
    wp = torch.zeros(128, 208, device="cuda", dtype=torch.half); wp[:, :196] = wgt.half()
    def timed(fn):
        fn(); torch.cuda.synchronize(); tt = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); tt.append(e0.elapsed_time(e1))
        return float(np.median(tt)), r
    t_f, o_f = timed(lambda: db.corr_pyramid_lookup_corr0(pyr, coords, wpk, bias))
    t_l, smp = timed(lambda: db.corr_pyramid_lookup(pyr, coords))
    t_c, o_u = timed(lambda: db.corr0_nchw(smp, wp, bias))
    d = (o_f.float() - o_u.float()).abs().max().item()
    print("fused lookup+corr0: %.3f ms   unfused: lookup %.3f + corr0_nchw %.3f = %.3f ms   max |diff| %.3e (scale %.2f)  reads+writes 744 B/ep -> %.1f GB/s" % (
        t_f, t_l, t_c, t_l + t_c, d, o_u.float().abs().max().item(), 744.0 * E * h * w / t_f / 1e6))
    for mode in ((2, 3, 5, 6) if db.get_option("ablation_build") else (6,)):      # 2 / 3 / 5 exist only in a -DDH_ABLATION build
        db.set_option("lookup_mode", mode)
        tm, _ = timed(lambda: db.corr_pyramid_lookup_corr0(pyr, coords, wpk, bias))
        print("  fused, variant %d (%s): %.3f ms" % (mode, {2: "no output stores", 3: "no tap loads", 5: "a quarter of the MFMAs", 6: "synchronous twin: every tap batch waited for at issue (correct results)"}[mode], tm))
    db.set_option("lookup_mode", 0)
    # refill schedule of the tap registers (lookup_fill 0 = by half level, 1 = by window row), alternating in one process
    for rep in range(3 if db.get_option("ablation_build") else 0):       # (lookup_fill 1 exists in -DDH_ABLATION builds only)
        for fill in (0, 1):
            db.set_option("lookup_fill", fill)
            tm, o_x = timed(lambda: db.corr_pyramid_lookup_corr0(pyr, coords, wpk, bias))
            line = "  lookup_fill %d: %.3f ms (744 B/ep -> %.1f GB/s)  identical to the first launch: %s" % (fill, tm, 744.0 * E * h * w / tm / 1e6, bool(torch.equal(o_x, o_f)))
            if db.get_option("ablation_build") and rep == 0:
                db.set_option("lookup_mode", 7)
                t7, _ = timed(lambda: db.corr_pyramid_lookup_corr0(pyr, coords, wpk, bias))
                db.set_option("lookup_mode", 0)
                line += "   [variant 7, taps loaded but neither interpolated nor multiplied: %.3f ms]" % t7
            print(line)
    if db.get_option("ablation_build"):
        db.set_option("lookup_fill", 0)
    for rep in range(2):
        t_f2, _ = timed(lambda: db.corr_pyramid_lookup_corr0(pyr, coords, wpk, bias))
        t_l2, _ = timed(lambda: db.corr_pyramid_lookup(pyr, coords))
        print("  again, product kernels: fused %.3f ms, unfused lookup %.3f ms" % (t_f2, t_l2))
