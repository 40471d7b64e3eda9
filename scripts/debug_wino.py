#!/usr/bin/env python
"""GPU debugging aid for the Winograd prototype: single-tap weights -> the output is a shifted copy of the input; prints where
the kernel's output differs (by image row, column parity, cout block, tap)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import numpy as np, torch
import droid_backends as db
from droid_amd.update import pack_conv, pack_conv_wino, EPI_LINEAR, LAYOUT_WINO
torch.manual_seed(0)
N, H, C, CO = 2, 8, 128, 128
x = torch.randn(N, H, 64, C, device="cuda").half()
for (dy, dx) in ((1, 1), (1, 0), (1, 2), (0, 1), (2, 1)):
    w = torch.zeros(CO, C, 3, 3, device="cuda")
    w[torch.arange(CO), torch.arange(C), dy, dx] = 1.0
    wp, bp = pack_conv(w, torch.zeros(CO, device="cuda"))
    out = torch.empty(N, H, 64, CO, device="cuda", dtype=torch.float16)
    db.conv2d_nhwc([x], wp, pack_conv_wino(w), bp, 3, 3, CO, EPI_LINEAR, out, CO, None, None, None, None, weights_layout=LAYOUT_WINO)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    e = (out.float() - ref).abs()
    print("tap (dy=%d, dx=%d): max err %.3f" % (dy, dx, e.max().item()))
    if e.max().item() > 0.01:
        print("  by image row :", np.round(e.amax(dim=(0, 2, 3)).cpu().numpy(), 2))
        print("  by col parity:", np.round([e[:, :, 0::2].max().item(), e[:, :, 1::2].max().item()], 2))
        print("  by column    :", np.round(e.amax(dim=(0, 1, 3)).cpu().numpy()[:16], 2))
        print("  by cout/32   :", np.round([e[..., 32 * k:32 * k + 32].max().item() for k in range(CO // 32)], 2))
        print("  by cout%8    :", np.round([e[..., k::8].max().item() for k in range(8)], 2))
        # does the output equal the input at some other shift / channel?
        o = out[0, 3].float(); best = None
        for sy in (-1, 0, 1):
            for sx in (-2, -1, 0, 1, 2):
                r = torch.roll(x[0].float(), shifts=(-sy, -sx), dims=(0, 1))[3]
                d = (o[4:60] - r[4:60]).abs().max().item()
                if best is None or d < best[0]: best = (d, sy, sx)
        print("  closest shifted input (row 3): err %.3f at (dy=%d, dx=%d)" % best)
