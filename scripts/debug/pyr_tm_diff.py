import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import torch
import droid_backends as db
torch.manual_seed(0)
h, w = 48, 64
f1 = torch.randn(1, 128, h, w, device="cuda").half(); f2 = torch.randn(1, 128, h, w, device="cuda").half()
db.set_option("pyr_build_tm", 0); a = db.corr_pyramid_build(f1, f2).clone()
db.set_option("pyr_build_tm", 1); b = db.corr_pyramid_build(f1, f2).clone()
torch.cuda.synchronize()
bad = torch.nonzero(a[0] != b[0])[:, 0]
print("differing halves:", len(bad), "of", a.numel())
blk = (h + 1) * w * 64                      # elements of one source block at level 0
L0 = 48 * blk
print("in level 0:", int((bad < L0).sum()), " beyond:", int((bad >= L0).sum()))
import collections
cnt = collections.Counter()
for o in bad[:200000:97].tolist():
    if o >= L0: continue
    sb, r = divmod(o, blk); v, r = divmod(r, w * 64); up, r = divmod(r, 128); p, par = divmod(r, 2)
    cnt[(v < 48, p >> 4, (2 * up + par) // 16)] += 1
print("(row valid, row pair of p, u // 16) -> count:", sorted(cnt.items())[:40])
o = bad[0].item(); sb, r = divmod(o, blk); v, r = divmod(r, w * 64); up, r = divmod(r, 128); p, par = divmod(r, 2)
print("first: sb %d v %d up %d p %d par %d: old %g new %g" % (sb, v, up, p, par, a[0][o].item(), b[0][o].item()))
# where does the wrong value come from?  all-pairs correlation of source pixel (sb, p) in fp32
F1 = f1[0].float().reshape(128, h * w); F2 = f2[0].float().reshape(128, h * w)
for o in bad[:5000:700].tolist():
    sb, r = divmod(o, blk); v, r = divmod(r, w * 64); up, r = divmod(r, 128); p, par = divmod(r, 2)
    by, bx = divmod(sb, 8); y1 = by * 8 + (p >> 3); x1 = bx * 8 + (p & 7)
    c = (F1[:, y1 * w + x1] @ F2 / 16).reshape(h, w)
    want_y, want_x = (y1 + v) % h, (x1 + 2 * up + par) % w
    got = b[0][o].float()
    cand = torch.nonzero((c - got).abs() < 2e-3)
    print("sb %d p %d (y1 %d x1 %d) v %d u %d: expected target (%d,%d) = %.4f [old %.4f]; new %.4f matches targets %s" % (
        sb, p, y1, x1, v, 2 * up + par, want_y, want_x, c[want_y, want_x].item(), a[0][o].item(), got.item(), cand.tolist()[:4]))
