import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "droid-slam_amd"))
import droid_backends as db
from droid_amd import synthetic as syn
from oracle import ba as oba
d = lambda a: torch.as_tensor(a).cuda().contiguous()
for seed in (5, 21, 4, 2, 33):
    for n in (6, 10):
        g = syn.small_graph(n_frames=n, seed=seed, ht=12, wd=16)
        for itrs in (1, 2):
            poses, disps = d(g["poses"]), d(g["disps"])
            dx, dz = db.ba(poses, disps, d(g["intrinsics"]), d(g["disps_sens"]), d(g["targets"]), d(g["weights"]),
                           d(g["eta"]), d(g["ii"]), d(g["jj"]), 1, n, itrs, 1e-4, 0.1, False)
            res = {}
            for name, dt in (("f64", np.float64), ("f32", np.float32)):
                rp = g["poses"].astype(np.float64); rd = g["disps"].astype(np.float64)
                rdx, rdz, info = oba.ba(rp, rd, g["intrinsics"], g["disps_sens"], g["targets"], g["weights"], g["eta"],
                                  g["ii"], g["jj"], 1, n, itrs, 1e-4, 0.1, False, dtype=dt, return_system=True)
                res[name] = (rdx, rdz, rp, rd, info)
            e64 = np.linalg.norm(dx.cpu().numpy() - res["f64"][0]) / np.linalg.norm(res["f64"][0])
            e32 = np.linalg.norm(res["f32"][0] - res["f64"][0]) / np.linalg.norm(res["f64"][0])
            H = res["f64"][4]["H"]; ev = np.linalg.eigvalsh(H + np.eye(len(H)) * 0.1)
            print("seed %d n %d itrs %d: gpu-vs-f64 %.2e  f32oracle-vs-f64 %.2e  |dx| %.3e cond %.2e  dposes %.2e ddisps %.2e" % (
                seed, n, itrs, e64, e32, np.linalg.norm(res["f64"][0]), ev[-1] / ev[0],
                np.abs(poses.cpu().numpy() - res["f64"][2]).max(), np.abs(disps.cpu().numpy() - res["f64"][3]).max()))
