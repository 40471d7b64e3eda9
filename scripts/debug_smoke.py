import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import numpy as np, torch
import droid_backends as db
from droid_amd import synthetic as syn
from oracle import ba as oba
g = syn.small_graph(n_frames=6, seed=21, ht=12, wd=16)
d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda().contiguous()
kx = np.unique(np.concatenate([np.arange(1, 6), g["ii"]]))
rng = np.random.default_rng(99)
eta2 = (0.2 * rng.uniform(1e-6, 1e-3, (len(kx),) + g["disps"].shape[1:]) + 1e-7).astype(np.float32)
for name, eta in (("graph eta", g["eta"]), ("test eta", eta2)):
    poses, disps = d(g["poses"]), d(g["disps"])
    dx, dz = db.ba(poses, disps, d(g["intrinsics"]), d(g["disps_sens"]), d(g["targets"]), d(g["weights"]),
                   d(eta), d(g["ii"]), d(g["jj"]), 1, 6, 1, 1e-4, 0.1, False)
    torch.cuda.synchronize()
    rp = g["poses"].astype(np.float64).copy(); rd = g["disps"].astype(np.float64).copy()
    rdx, rdz = oba.ba(rp, rd, g["intrinsics"], g["disps_sens"], g["targets"], g["weights"], eta, g["ii"], g["jj"], 1, 6, 1, 1e-4, 0.1, False)
    err = np.abs(disps.cpu().numpy() - rd) / np.maximum(1.0, np.abs(rd))
    print(name, "eta shape", eta.shape, eta.dtype, "dx rel", np.linalg.norm(dx.cpu().numpy() - rdx) / np.linalg.norm(rdx),
          "q995", np.quantile(err, 0.995), "max", err.max(), "per-frame max", err.reshape(6, -1).max(1),
          "dz shapes", dz.shape, rdz.shape, "dz err", np.abs(dz.cpu().numpy() - rdz).max())
