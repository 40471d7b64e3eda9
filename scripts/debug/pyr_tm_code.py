import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import torch
import droid_backends as db
h, w = 48, 64
yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
def build(src_code, tgt_code, tm):
    f1 = torch.zeros(1, 128, h, w); f2 = torch.zeros(1, 128, h, w)
    f1[0, 0] = src_code * 16.0; f1[0, 1] = 16.0; f2[0, 0] = 1.0; f2[0, 1] = tgt_code     # corr / 16 = src_code + tgt_code
    db.set_option("pyr_build_tm", 0); db.set_option("pyr_build_waves", 8 if tm else 4)      # (here: A = 4 waves, B = 8 waves of the same roles)
    return db.corr_pyramid_build(f1.cuda().half(), f2.cuda().half())[0].float().cpu()
blk = (h + 1) * w * 64
res = {}
for name, sc, tc in (("y2", 0 * yy, yy.float()), ("x2", 0 * xx, xx.float()), ("y1", yy.float(), 0 * yy), ("x1", xx.float(), 0 * xx)):
    res[name] = (build(sc, tc, 0)[:48 * blk], build(sc, tc, 1)[:48 * blk])
bad = torch.nonzero((res["y2"][0] != res["y2"][1]) | (res["x2"][0] != res["x2"][1]) | (res["y1"][0] != res["y1"][1]) | (res["x1"][0] != res["x1"][1]))[:, 0]
print("cells differing:", len(bad))
import collections
lev0 = 48 * blk
cnt = collections.Counter()
for o in bad.tolist()[::53]:
    sb, r = divmod(o, blk); v, r = divmod(r, w * 64); up, r = divmod(r, 128); p, par = divmod(r, 2)
    cnt[("v", v)] += 1; cnt[("sb", sb)] += 1; cnt[("u16", (2 * up + par) // 16)] += 1; cnt[("prow", p >> 3)] += 1
for key in ("v", "sb", "u16", "prow"):
    print(key, sorted((k[1], c) for k, c in cnt.items() if k[0] == key))
for o in bad[::max(1, len(bad) // 24)].tolist()[:24]:
    sb, r = divmod(o, blk); v, r = divmod(r, w * 64); up, r = divmod(r, 128); p, par = divmod(r, 2)
    print("sb %2d v %2d u %2d p %2d | should hold (y1 %d x1 %d y2 %d x2 %d) | holds (y1 %d x1 %d y2 %d x2 %d)" % (
        sb, v, 2 * up + par, p, res["y1"][0][o], res["x1"][0][o], res["y2"][0][o], res["x2"][0][o],
        res["y1"][1][o], res["x1"][1][o], res["y2"][1][o], res["x2"][1][o]))
