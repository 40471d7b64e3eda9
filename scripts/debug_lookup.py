import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import numpy as np, torch
import droid_backends as db
from oracle import corr as ocorr
E, h, w = 3, 16, 16
rng = np.random.default_rng(E * 1000 + h + w)
f1 = rng.standard_normal((E, 128, h, w)).astype(np.float16)
f2 = rng.standard_normal((E, 128, h, w)).astype(np.float16)
yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
c = np.zeros((E, h, w, 2))
for e in range(E):
    a = rng.uniform(-6, 6, 6)
    c[e, ..., 0] = xx + a[0] + a[1] * xx / w + a[2] * yy / h
    c[e, ..., 1] = yy + a[3] + a[4] * xx / w + a[5] * yy / h
c = c.astype(np.float32)
d = lambda a: torch.as_tensor(a).cuda().contiguous()
pyr = db.corr_pyramid_build(d(f1), d(f2))
for pf in ("1", "0"):
    os.environ["DH_LOOKUP_PREFETCH"] = pf
    out = db.corr_pyramid_lookup(pyr, d(c)).float().cpu().numpy()
    ref = ocorr.corr_block_lookup(ocorr.corr_pyramid(f1, f2, 4), c, 3)
    err = np.abs(out - ref)
    tol = 2.0 ** -8 * np.abs(ref).max()
    bad = err > tol
    print("prefetch", pf, "max err", err.max(), "tol", tol, "bad frac", bad.mean())
    o = bad.reshape(E, 4, 7, 7, h, w)
    print(" by level", o.mean(axis=(0, 2, 3, 4, 5)))
    print(" by a(x off)", o.mean(axis=(0, 1, 3, 4, 5)))
    print(" by b(y off)", o.mean(axis=(0, 1, 2, 4, 5)))
    print(" by x parity", o[..., 0::2].mean(), o[..., 1::2].mean())
    print(" by row", o.mean(axis=(0, 1, 2, 3, 5)))
    print(" by col", o.mean(axis=(0, 1, 2, 3, 4)))
