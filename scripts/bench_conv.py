#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM convolution shapes of the update operator (one MI355X)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# DH_ABLATION_BUILD=1: the -DDH_ABLATION library (droid-slam_amd/ablation/), for the kernels that only exist there
sys.path[:0] = [ROOT] + ([os.path.join(ROOT, "droid-slam_amd", "ablation")] if os.environ.get("DH_ABLATION_BUILD") else []) + [os.path.join(ROOT, "droid-slam_amd")]
import torch
import droid_backends as db
from droid_amd.update import pack_conv, pack_conv_halo, pack_conv_wino, EPI_RELU, EPI_LINEAR, LAYOUT_WINO
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h, w = 48, 64
torch.manual_seed(0)
shapes = [("gru z|r 3x3 448->256", (128, 128, 128, 64), 256, 3), ("gates z|r 3x3 320->256", (128, 128, 64), 256, 3), ("gru q 3x3 448->128", (128, 128, 128, 64), 128, 3),
          ("3x3 128->128", (128,), 128, 3), ("heads0 3x3 128->256", (128,), 256, 3), ("heads2 3x3 256->4", (256,), 4, 3),
          ("corr0 1x1 200->128", (200,), 128, 1), ("flow0 7x7 8->128", (8,), 128, 7), ("flow2 3x3 128->64", (128,), 64, 3),
          ("upmask 1x1 128->576 (K=512 frames)", (128,), 576, 1)]
only = os.environ.get("DH_SHAPE")        # substring filter (profiling one shape)
reps = int(os.environ.get("DH_REPS", "3"))
for name, cins, cout, k in shapes:
    if only and only not in name:
        continue
    n = E // 8 if "upmask" in name else E
    fill = os.environ.get("DH_FILL", "randn")       # "zero": operands of zeros (DVFS headroom check, MI355X_MICROARCH.md)
    xs = [(torch.zeros if fill == "zero" else torch.randn)(n, h, w, c, device="cuda").half() for c in cins]
    wgt = (torch.zeros if fill == "zero" else torch.randn)(cout, sum(cins), k, k, device="cuda") / (sum(cins) * k * k) ** 0.5
    wp, bp = pack_conv(wgt, torch.zeros(cout, device="cuda"))
    wh = pack_conv_halo(wgt) if not os.environ.get("DH_CONV_NO_HALO") else None
    out = torch.empty(n, h, w, cout, device="cuda", dtype=torch.float16)
    epi = EPI_LINEAR if cout <= 4 else EPI_RELU
    run = lambda: db.conv2d_nhwc(xs, wp, wh, bp, k, k, cout, epi, out, cout, None, None, None, None)
    run(); torch.cuda.synchronize()
    if os.environ.get("DH_CHECK"):          # numerics of the first two images against torch's fp32 convolution
        x = torch.cat([t[:2] for t in xs], -1).float().permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(x, wgt.half().float(), None, padding=k // 2).permute(0, 2, 3, 1)
        ref = ref if epi == EPI_LINEAR else ref.clamp_min(0)
        print("   max |err| = %.4f (ref max %.2f)" % ((out[:2].float() - ref).abs().max().item(), ref.abs().max().item()))
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = min(ts); fl = 2.0 * n * h * w * sum(cins) * k * k * cout
    print("%-38s %8.3f ms  %7.1f TFLOP/s (useful)  %6.1f%% of 2.5 PF" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / 25.0))
    ww = pack_conv_wino(wgt) if (k == 3 and os.environ.get("DH_WINO")) else None
    if ww is not None:                      # the Winograd F(2,3) prototype on the same operands (DH_WINO=1)
        out2 = torch.empty_like(out)
        runw = lambda: db.conv2d_nhwc(xs, wp, ww, bp, k, k, cout, epi, out2, cout, None, None, None, None, weights_layout=LAYOUT_WINO)
        runw(); torch.cuda.synchronize()
        x = torch.cat([t[:2] for t in xs], -1).float().permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(x, wgt.half().float(), None, padding=1).permute(0, 2, 3, 1).clamp_min(0)
        e_dir = (out[:2].float() - ref).abs().max().item(); e_win = (out2[:2].float() - ref).abs().max().item()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); runw(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        msw = min(ts)
        print("%-38s %8.3f ms  %7.1f TFLOP/s (useful)  %6.1f%% of 2.5 PF   winograd F(2,3)x: x%.2f vs direct; max |err| vs fp32 conv: direct %.4f, winograd %.4f (|ref| max %.2f)" % (
            "  -> winograd prototype", msw, fl / msw / 1e9, fl / msw / 1e9 / 25.0, ms / msw, e_dir, e_win, ref.abs().max().item()))
