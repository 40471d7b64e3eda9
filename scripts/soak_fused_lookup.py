#!/usr/bin/env python
"""Soak test of the fused lookup's explicit vmcnt bookkeeping: many launches on different flows (and under concurrent load on a
second stream), each compared BIT FOR BIT with the synchronous twin (lookup_mode 6).  usage: python scripts/soak_fused_lookup.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import numpy as np, torch
import droid_backends as db
from droid_amd.update import pack_corr0_fused

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
E, h, w = 1024, 48, 64
torch.manual_seed(0)
f = torch.randn(64, 128, h, w, device="cuda").half()
i1 = torch.randint(0, 64, (E,), device="cuda"); i2 = torch.randint(0, 64, (E,), device="cuda")
pyr = db.corr_pyramid_build(f[i1].contiguous(), f[i2].contiguous())
wgt = torch.randn(128, 196, device="cuda") * 0.05
bias = torch.randn(128, device="cuda") * 0.3
wpk = pack_corr0_fused(wgt)
yy, xx = torch.meshgrid(torch.arange(h, device="cuda", dtype=torch.float32), torch.arange(w, device="cuda", dtype=torch.float32), indexing="ij")
side = torch.cuda.Stream()
noise_a = torch.randn(4096, 4096, device="cuda").half(); noise_b = torch.randn(4096, 4096, device="cuda").half()
bad = 0
for it in range(iters):
    g = torch.Generator(device="cuda"); g.manual_seed(it)
    am = (torch.rand(E, 6, 1, 1, device="cuda", generator=g) - 0.5) * (4.0 + it % 9)
    c = torch.stack([xx + am[:, 0] + am[:, 1] * xx / w + am[:, 2] * yy / h, yy + am[:, 3] + am[:, 4] * xx / w + am[:, 5] * yy / h], -1).contiguous()
    if it % 5 == 4:
        c = c + torch.rand(E, h, w, 2, device="cuda", generator=g) * 3.0                      # incoherent component
    db.set_option("lookup_mode", 6)
    ref = db.corr_pyramid_lookup_corr0(pyr, c, wpk, bias)
    db.set_option("lookup_mode", 0)
    torch.cuda.synchronize()
    if it % 2:
        with torch.cuda.stream(side):                     # a GEMM competing for the CUs / HBM while the kernel runs
            for _ in range(4):
                noise_a @ noise_b
    out = db.corr_pyramid_lookup_corr0(pyr, c, wpk, bias)
    torch.cuda.synchronize()
    if not torch.equal(out, ref):
        bad += 1
        print("iteration %d: %d of %d values differ" % (it, int((out != ref).sum()), out.numel()))
print("soak: %d iterations, %d mismatching launches" % (iters, bad))
sys.exit(1 if bad else 0)
