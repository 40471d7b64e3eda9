#!/usr/bin/env python
"""Micro-benchmark of altcorr_forward (on-the-fly correlation, reference altcorr_kernel.cu) at the 4 pyramid levels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import numpy as np, torch
import droid_backends as db
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, C, H, W = 64, 128, 48, 64
torch.manual_seed(0)
f1 = torch.randn(1, N, C, H, W, device="cuda").half()
ii = torch.randint(0, N, (M,), device="cuda"); jj = torch.randint(0, N, (M,), device="cuda")
yy, xx = torch.meshgrid(torch.arange(H, device="cuda", dtype=torch.float32), torch.arange(W, device="cuda", dtype=torch.float32), indexing="ij")
coords = torch.stack([xx + 1.7, yy - 2.3], 0)[None, None].repeat(1, M, 1, 1, 1).contiguous()
tot = 0.0
for l in range(4):
    f2 = torch.nn.functional.avg_pool2d(f1[0].float(), 2 ** l, stride=2 ** l).half()[None] if l else f1
    c = (coords / 2 ** l).contiguous()
    if len(sys.argv) > 2 and sys.argv[2] == "mfma":
        a1 = f1[0].permute(0, 2, 3, 1).contiguous(); a2 = f2[0].permute(0, 2, 3, 1).contiguous(); c1 = c[0].contiguous()
        run = lambda: db.altcorr_forward_nhwc(a1, a2, c1, ii, jj)
    else:
        run = lambda: db.altcorr_forward(f1, f2.contiguous(), c, ii, jj, 3)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = min(ts); tot += ms
    fl = 2.0 * M * H * W * 64 * C
    print("level %d: %.3f ms  %.1f TFLOP/s" % (l, ms, fl / ms / 1e9))
print("4 levels, %d edges: %.3f ms -> %.1f M edge-pixels/s" % (M, tot, M * H * W / tot / 1e3))
