import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import torch, droid_backends as db
torch.manual_seed(0)
h, w = 48, 64
f1 = torch.randn(1, 128, h, w, device="cuda").half(); f2 = torch.randn(1, 128, h, w, device="cuda").half()
outs = {}
for tm, waves in ((0, 4), (0, 8), (1, 8)):
    db.set_option("pyr_build_tm", tm); db.set_option("pyr_build_waves", waves)
    outs[(tm, waves)] = db.corr_pyramid_build(f1, f2)[0].clone()
torch.cuda.synchronize()
ref = outs[(0, 4)]
off = 0
for l in range(4):
    h2, w2 = h >> l, w >> l
    blk = (h2 + 1) * w2 * 64
    n = 48 * blk
    for key in ((0, 8), (1, 8)):
        d = (outs[key][off:off + n] != ref[off:off + n])
        idx = torch.nonzero(d)[:, 0]
        msg = ""
        if len(idx):
            o = idx[0].item(); sb, r = divmod(o, blk); v, r = divmod(r, w2 * 64); up, r = divmod(r, 128); p, par = divmod(r, 2)
            vs = torch.unique(((idx % blk) // (w2 * 64)))[:20].tolist()
            ps = torch.unique(((idx % (w2 * 64)) % 128) // 2)[:70].tolist()
            ups = torch.unique(((idx % (w2 * 64)) // 128))[:40].tolist()
            msg = "first sb %d v %d up %d p %d; rows v: %s; cell pairs up: %s; pixels p: %s" % (sb, v, up, p, vs, ups, ps)
        print("level %d %s: %d of %d halves differ. %s" % (l, key, len(idx), n, msg))
    off += n
