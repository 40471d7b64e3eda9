import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import numpy as np, torch
from droid_amd import synthetic as syn
from droid_amd.corr import CorrBlock
from droid_amd.depth_video import DepthVideo
g = syn.make_graph("C3", with_features=True)
N, ht, wd = g["n_frames"], g["ht"], g["wd"]
dev = "cuda:0"
d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)
video = DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, stereo=False, device=dev)
video.fmaps[:N] = d(g["fmaps"])
order = np.argsort(g["ii"], kind="stable")
ii, jj = d(g["ii"][order]), d(g["jj"][order])
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    return ts
arena = CorrBlock.arena(len(ii), ht, wd, dev)
print("video.fmaps", tuple(video.fmaps.shape), video.fmaps.dtype, video.fmaps.is_contiguous())
print("from_frames(video.fmaps, out=arena) ms:", t(lambda: CorrBlock.from_frames(video.fmaps, ii, jj, out=arena)))
fm = d(g["fmaps"])
print("from_frames(plain fmaps, out=arena) ms:", t(lambda: CorrBlock.from_frames(fm, ii, jj, out=arena)))
c = torch.zeros_like(ii)
print("CorrBlock(gathered, out=arena) ms:", t(lambda: CorrBlock(video.fmaps[ii, 0][None], video.fmaps[jj, c][None], out=arena)))
def parts():
    Nn, rig, C, h, w = video.fmaps.shape
    t0 = time.perf_counter(); fr, inv = torch.unique(torch.cat([ii * rig, jj * rig]), return_inverse=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    f = video.fmaps.reshape(Nn * rig, C, h, w)[fr].half().contiguous(); torch.cuda.synchronize(); t2 = time.perf_counter()
    import droid_backends as db
    prep = db.corr_pyramid_prepare_frames(f, h, w); torch.cuda.synchronize(); t3 = time.perf_counter()
    p = db.corr_pyramid_build_indexed(prep, inv[:len(ii)].contiguous(), inv[len(ii):].contiguous(), h, w, arena); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("  unique %.2f gather %.2f prepare %.2f build %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3)))
parts(); parts()
