import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from droid_amd.corr import CorrBlock
from droid_amd.update import corr_channel_map
torch.manual_seed(5)
for (E, h, w) in [(3, 16, 16), (2, 48, 64)]:
    f1 = torch.randn(1, E, 128, h, w, device="cuda").half()
    f2 = torch.randn(1, E, 128, h, w, device="cuda").half()
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    c = np.zeros((E, h, w, 2))
    rng = np.random.default_rng(2)
    for e in range(E):
        a6 = rng.uniform(-6, 6, 6)
        c[e, ..., 0] = xx + a6[0] + a6[1] * xx / w + a6[2] * yy / h
        c[e, ..., 1] = yy + a6[3] + a6[4] * xx / w + a6[5] * yy / h
    coords = torch.as_tensor(c.astype(np.float32)).cuda()[None]
    blk = CorrBlock(f1, f2)
    a = blk(coords)[0].permute(0, 2, 3, 1)
    b = blk.lookup_nhwc(coords).permute(1, 2, 3, 0, 4).reshape(E, h, w, 224)
    m = corr_channel_map().cuda()
    bb = b[..., m >= 0].float(); aa = a[..., m[m >= 0]].float()
    bad = (bb != aa)
    print((E, h, w), "bad frac", bad.float().mean().item(), "max diff", (bb - aa).abs().max().item(), "pads nonzero", torch.count_nonzero(b[..., m < 0]).item())
    if bad.any():
        idx = bad.nonzero()
        print(" first bad", idx[:5].tolist())
        print(" bad by level", [bad.reshape(E, h, w, 4, 49)[..., l, :].float().mean().item() for l in range(4)])
        print(" bad by chan-in-level", bad.reshape(E, h, w, 4, 49).float().mean((0, 1, 2, 3)).tolist())
        print(" bad by x", bad.float().mean((0, 1, 3)).tolist())
