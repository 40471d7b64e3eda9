#!/usr/bin/env python
"""Generate tests/golden/ref_cuda.npz: outputs of the REFERENCE's own kernels (src/droid_kernels.cu,
correlation_kernels.cu, altcorr_kernel.cu compiled for gfx950 by oracle/build_ref.py) on small seeded inputs.

Runs on a GPU box only (the reference has no CPU build):

    gpurun -- 'python tests/golden/make_ref_golden.py gpurun_out/golden/ref_cuda.npz'

and the result is copied to tests/golden/ref_cuda.npz and committed.  tests/test_oracle_golden.py pins the CPU oracle
against these vectors, so the CUDA-only semantics (damping placement, MIN_DEPTH 0.25, EvT6x1 row skip, fp16 lookup
accumulation, geometry kernels) are pinned by reference-produced numbers, not by reading alone.
Inputs come from droid_amd.synthetic (seeded numpy) and are stored next to the outputs.
"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "droid-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import build_ref          # noqa: E402
from droid_amd import synthetic as syn  # noqa: E402


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def eta_for(g, t0, t1, seed=99):
    kx = np.unique(np.concatenate([np.arange(t0, t1), g["ii"]]))
    rng = np.random.default_rng(seed)
    return (0.2 * rng.uniform(1e-6, 1e-3, (len(kx),) + g["disps"].shape[1:]) + 1e-7).astype(np.float32)


BA_CASES = {
    # name: (small_graph kwargs, t0, lm, ep, motion_only)
    "mono": (dict(n_frames=6, seed=21, ht=12, wd=16), 1, 1e-4, 0.1, False),
    "stereo": (dict(n_frames=6, seed=21, ht=12, wd=16, stereo=True), 1, 1e-4, 0.1, False),
    "sensor": (dict(n_frames=6, seed=21, ht=12, wd=16, sensor_depth=True), 1, 1e-4, 0.1, False),
    "t0_3": (dict(n_frames=6, seed=21, ht=12, wd=16), 3, 1e-4, 0.1, False),
    "global": (dict(n_frames=7, seed=5, ht=12, wd=16, radius=3), 1, 1e-5, 1e-2, False),
    "motion": (dict(n_frames=6, seed=4, ht=12, wd=16), 1, 1e-4, 0.1, True),
}


def main(out_path):
    loaded = build_ref.load()
    assert loaded is not None, "oracle/_ref/droid_backends_ref.so missing: run oracle/build_ref.py where /root/reference exists"
    ref, ops = loaded
    out = {}
    for name, (kw, t0, lm, ep, mo) in BA_CASES.items():
        g = syn.small_graph(**kw)
        N = g["n_frames"]
        eta = eta_for(g, t0, N)
        for k in ("poses", "disps", "intrinsics", "disps_sens", "targets", "weights", "ii", "jj"):
            out["ba_%s_%s" % (name, k)] = g[k]
        out["ba_%s_eta" % name] = eta
        out["ba_%s_args" % name] = np.array([t0, N, lm, ep, float(mo)], dtype=np.float64)
        args = [dev(g[k]) for k in ("intrinsics", "disps_sens", "targets", "weights")] + [dev(eta), dev(g["ii"]), dev(g["jj"])]
        for itrs in (1, 2):
            poses, disps = dev(g["poses"]), dev(g["disps"])
            r = ref.ba(poses, disps, *args, t0, N, itrs, lm, ep, mo)
            torch.cuda.synchronize()
            out["ba_%s_poses%d" % (name, itrs)] = poses.cpu().numpy()
            out["ba_%s_disps%d" % (name, itrs)] = disps.cpu().numpy()
            out["ba_%s_dx%d" % (name, itrs)] = r[0].cpu().numpy()
            if not mo:
                out["ba_%s_dz%d" % (name, itrs)] = r[1].cpu().numpy()
        blk = ops.edge_blocks(dev(g["poses"]), dev(g["disps"]), args[0], args[2], args[3], args[5], args[6])
        for nm, t in zip(("Hs", "vs", "Eii", "Eij", "Cii", "bz"), blk):
            out["ba_%s_%s" % (name, nm)] = t.cpu().numpy()
        sysm = ops.reduced_system(dev(g["poses"]), dev(g["disps"]), *args, t0, N, mo)
        out["ba_%s_H" % name] = sysm[0].cpu().numpy()
        out["ba_%s_b" % name] = sysm[1].cpu().numpy()
        if not mo:
            out["ba_%s_C" % name] = sysm[2].cpu().numpy()
            out["ba_%s_w" % name] = sysm[3].cpu().numpy()

    # Cholesky failure -> zero update (droid_kernels.cu:1211-1219)
    g = syn.small_graph(n_frames=5, seed=2, ht=12, wd=16)
    poses, disps = dev(g["poses"]), dev(g["disps"])
    r = ref.ba(poses, disps, dev(g["intrinsics"]), dev(g["disps_sens"]), dev(g["targets"]), dev(g["weights"]),
               dev(g["eta"]), dev(g["ii"]), dev(g["jj"]), 1, 5, 1, 0.0, -1e9, True)
    out["fail_dx"] = r[0].cpu().numpy()
    out["fail_poses_unchanged"] = np.array(np.array_equal(poses.cpu().numpy(), g["poses"]))

    # correlation lookup (correlation_kernels.cu:20-71): fp32 and fp16 (accumulated in fp16 in global memory)
    rng = np.random.default_rng(17)
    shape = (2, 6, 8, 12, 16)
    vol = rng.standard_normal(shape).astype(np.float32)
    x = rng.uniform(-4, 16 - 1 + 4, (2, 6, 8)); y = rng.uniform(-4, 12 - 1 + 4, (2, 6, 8))
    x[:, 0, 0] = 5.0; y[:, 0, 0] = 2.0; x[:, -1, -1] = -50.0; y[:, -1, 0] = 1e4
    coords = np.stack([x, y], 1).astype(np.float32)
    out["ci_vol"] = vol.astype(np.float16)              # fp16-representable values for both dtypes
    out["ci_coords"] = coords
    v16 = dev(out["ci_vol"])
    out["ci_out_f32"] = ref.corr_index_forward(v16.float(), dev(coords), 3)[0].cpu().numpy()
    out["ci_out_f16"] = ref.corr_index_forward(v16, dev(coords), 3)[0].cpu().numpy()
    gcorr = rng.standard_normal((2, 7, 7, 6, 8)).astype(np.float32)
    out["ci_grad"] = gcorr
    out["ci_vgrad"] = ref.corr_index_backward(v16.float(), dev(coords), dev(gcorr), 3)[0].cpu().numpy()

    # alt correlation (altcorr_kernel.cu:24-75,132-172) incl. the permuted output view, two pyramid levels
    B, N, C, H, W = 1, 4, 32, 12, 16
    fm = rng.standard_normal((B, N, C, H, W)).astype(np.float16)
    ii = np.array([0, 1, 3, 2, 2]); jj = np.array([1, 0, 3, 0, 3])
    out["alt_fmap"] = fm; out["alt_ii"] = ii; out["alt_jj"] = jj
    f2 = torch.nn.functional.avg_pool2d(dev(fm).float().view(N, C, H, W), 2, 2).view(B, N, C, H // 2, W // 2)
    out["alt_fmap_l1"] = f2.half().cpu().numpy()
    for lvl, (fmap2, H2, W2) in enumerate(((dev(fm), H, W), (f2.half().contiguous(), H // 2, W // 2))):
        c = np.stack([rng.uniform(-3, W2 + 2, (B, 5, H, W)), rng.uniform(-3, H2 + 2, (B, 5, H, W))], 2).astype(np.float32)
        out["alt_coords_l%d" % lvl] = c
        o16 = ref.altcorr_forward(dev(fm), fmap2, dev(c), dev(ii), dev(jj), 3)[0]
        o32 = ref.altcorr_forward(dev(fm).float(), fmap2.float(), dev(c), dev(ii), dev(jj), 3)[0]
        out["alt_out_f16_l%d" % lvl] = o16.contiguous().cpu().numpy()
        out["alt_out_f32_l%d" % lvl] = o32.contiguous().cpu().numpy()
    # backward (altcorr_kernel.cu:78-129): float atomics -> order-dependent in the last bits only
    c0 = out["alt_coords_l0"]
    gal = rng.standard_normal((B, 5, 7, 7, H, W)).astype(np.float32)
    out["alt_grad"] = gal
    # binding order (droid.cpp:206-222): corr_grad is the 4th argument, given in the forward's output (x-outer) layout
    g1, g2 = ref.altcorr_backward(dev(fm).float(), dev(fm).float(), dev(c0), dev(gal), dev(ii), dev(jj), 3)
    out["alt_g1"] = g1.cpu().numpy(); out["alt_g2"] = g2.cpu().numpy()

    # geometry kernels (droid_kernels.cu:436-859)
    g = syn.small_graph(n_frames=8, seed=13, ht=12, wd=16)
    poses, disps, intr = dev(g["poses_gt"]), dev(g["disps_gt"]), dev(g["intrinsics"])
    out["geo_poses"] = g["poses_gt"]; out["geo_disps"] = g["disps_gt"]; out["geo_intr"] = g["intrinsics"]
    out["geo_ii"] = g["ii"]; out["geo_jj"] = g["jj"]
    out["geo_dist"] = ref.frame_distance(poses, disps, intr, dev(g["ii"]), dev(g["jj"]), 0.3).cpu().numpy()
    pm = ref.projmap(poses, disps, intr, dev(g["ii"]), dev(g["jj"]))
    out["geo_pm_coords"] = pm[0].cpu().numpy(); out["geo_pm_valid"] = pm[1].cpu().numpy()
    out["geo_points"] = ref.iproj(poses, disps, intr).cpu().numpy()
    ix = np.arange(8); th = np.full(8, 0.05, dtype=np.float32)
    out["geo_ix"] = ix; out["geo_th"] = th
    out["geo_count"] = ref.depth_filter(poses, disps, intr, dev(ix), dev(th)).cpu().numpy()

    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, "%d arrays, %.1f KB" % (len(out), os.path.getsize(out_path) / 1024))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden", "ref_cuda.npz"))
