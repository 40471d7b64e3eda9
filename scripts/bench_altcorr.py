#!/usr/bin/env python
"""Micro-benchmark of the on-the-fly correlation (reference altcorr_kernel.cu) at the 4 pyramid levels, coherent flow.
usage: python scripts/bench_altcorr.py [edges=512] [valu|mfma]
  valu  droid_backends.altcorr_forward (the reference's entry point, NCHW features)
  mfma  droid_backends.altcorr_forward_nhwc_levels (channel-last features, what droid_amd.corr.AltCorrBlock calls): second
        form of the MFMA kernel (LDS-DMA staging), then the first (DH_ALTCORR_V1) on the same inputs, and their difference"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import numpy as np, torch
import droid_backends as db
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mode = sys.argv[2] if len(sys.argv) > 2 else "mfma"
N, C, H, W = 64, 128, 48, 64
torch.manual_seed(0)
f1 = torch.randn(1, N, C, H, W, device="cuda").half()
ii = torch.randint(0, N, (M,), device="cuda"); jj = torch.randint(0, N, (M,), device="cuda")
yy, xx = torch.meshgrid(torch.arange(H, device="cuda", dtype=torch.float32), torch.arange(W, device="cuda", dtype=torch.float32), indexing="ij")
shift = torch.rand(M, 2, 1, 1, device="cuda") * 16 - 8
coords = (torch.stack([xx + 1.7, yy - 2.3], 0)[None] + shift).contiguous()            # [M,2,H,W], full resolution


def timed(run):
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)), out


flop = 65536.0 * M * H * W
if mode == "valu":
    tot = 0.0
    for l in range(4):
        f2 = torch.nn.functional.avg_pool2d(f1[0].float(), 2 ** l, stride=2 ** l).half()[None] if l else f1
        c = (coords[None] / 2 ** l).contiguous()
        ms, _ = timed(lambda: db.altcorr_forward(f1, f2.contiguous(), c, ii, jj, 3))
        tot += ms
        print("level %d: %.3f ms" % (l, ms))
    print("valu, 4 levels, %d edges: %.3f ms  %.1f TFLOP/s" % (M, tot, flop / tot / 1e9))
else:
    lv = [f1[0].permute(0, 2, 3, 1).contiguous()]
    for l in range(1, 4):
        lv.append(torch.nn.functional.avg_pool2d(f1[0].float(), 2 ** l, stride=2 ** l).half().permute(0, 2, 3, 1).contiguous())
    res = {}
    for v1 in (0, 1):
        db.set_option("altcorr_v1", v1)
        ms, out = timed(lambda: db.altcorr_forward_nhwc_levels(lv[0], lv, coords, ii, jj))
        res[v1] = out
        print("mfma %s form, 4 levels, %d edges: %.3f ms  %.1f TFLOP/s (%.1f%% of 2500)" % ("first" if v1 else "second", M, ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0))
    db.set_option("altcorr_v1", 0)
    print("max |second - first| = %g (max |out| %g)" % ((res[0].float() - res[1].float()).abs().max().item(), res[1].float().abs().max().item()))
