"""Pin the oracle against vectors produced by the REFERENCE's own Python code
(tests/golden/make_golden.py -> geom/ba.py, geom/projective_ops.py, modules/corr.py, droid_net.py)."""
import os
import numpy as np
import torch

from oracle import ba as oba, corr as ocorr, geom as ogeom, update as oupd
from droid_amd.weights import deterministic_state_dict


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_reprojection_and_jacobians_match_reference_python(golden_dir):
    g = _load(golden_dir, "ba_python.npz")
    E, _, ht, wd = g["targets"].shape
    coords, valid = ogeom.projective_transform(g["poses"], g["disps"], g["intrinsics"], g["ii"], g["jj"])
    assert np.abs(coords - g["coords"]).max() < 1e-7   # poses are fp32-rounded (|q|-1 ~ 1e-7): rel-pose composition order differs
    assert np.array_equal(valid, g["valid"])
    # analytic Jacobians: the CUDA-kernel form (oracle.ba.edge_terms) equals Jp@Ja of projective_ops.py
    # wherever both use the same depth threshold (Z >= 0.25)
    T = oba.edge_terms(g["poses"], g["disps"], g["intrinsics"], g["targets"], g["weights"], g["ii"], g["jj"])
    Jj_ref = np.moveaxis(g["Jj"].reshape(E, ht * wd, 2, 6), 1, -1)     # [E,2,6,HW]
    Ji_ref = np.moveaxis(g["Ji"].reshape(E, ht * wd, 2, 6), 1, -1)
    Jz_ref = np.moveaxis(g["Jz"].reshape(E, ht * wd, 2), 1, -1)         # [E,2,HW]
    ok = (T["w"].sum(1) > 0) | (g["weights"].reshape(E, 2, -1).sum(1) == 0)
    far = np.broadcast_to((np.abs(Jz_ref).sum(1) >= 0)[:, None, None], Jj_ref.shape)
    z_ok = np.ones((E, ht * wd), bool)
    # pixels with Z < 0.25 are zeroed by the kernel form; compare the rest
    zmask = np.abs(T["Jj"]).sum((1, 2)) > 0
    for name, mine, ref in (("Jj", T["Jj"], Jj_ref), ("Ji", T["Ji"], Ji_ref)):
        d = np.abs(mine - ref) * zmask[:, None, None]
        assert d.max() < 1e-6 * max(1.0, np.abs(ref).max()), name
    assert (np.abs(T["Jz"] - Jz_ref) * zmask[:, None]).max() < 1e-7
    assert zmask.mean() > 0.9


def test_dense_ba_step_matches_reference_python(golden_dir):
    g = _load(golden_dir, "ba_python.npz")
    p1, d1, dx, dz = oba.ba_dense_python_formulation(
        g["poses"], g["disps"], g["intrinsics"], g["targets"], g["weights"], g["eta"],
        g["ii"], g["jj"], fixedp=int(g["fixedp"]))
    assert np.abs(p1 - g["poses1"]).max() < 1e-8
    assert np.abs(d1 - g["disps1"]).max() < 1e-7
    p2, d2, _, _ = oba.ba_dense_python_formulation(
        p1, d1, g["intrinsics"], g["targets"], g["weights"], g["eta"], g["ii"], g["jj"], fixedp=int(g["fixedp"]))
    assert np.abs(p2 - g["poses2"]).max() < 1e-7
    assert np.abs(d2 - g["disps2"]).max() < 1e-6


def test_sparse_cuda_formulation_agrees_with_dense_when_semantics_coincide(golden_dir):
    """ba_cuda restatement vs the dense formulation: same linear system when the
    differing knobs (Q6 row skip, damping placement) are neutralised: lm=ep=0 is
    singular, so compare the assembled (A-S, b) instead of the solution."""
    g = _load(golden_dir, "ba_python.npz")
    N = g["poses"].shape[0]
    poses = g["poses"].copy(); disps = g["disps"].copy()
    ii, jj = g["ii"], g["jj"]
    kx = np.unique(np.concatenate([np.arange(1, N), ii]))
    kx_py = np.unique(ii)
    assert np.array_equal(kx, kx_py)
    dx, dz, info = oba.ba(poses, disps, g["intrinsics"], np.zeros_like(disps), g["targets"], g["weights"],
                          g["eta"], ii, jj, 1, N, 1, 0.0, 0.0, False, return_system=True)
    # rebuild the dense system from the same blocks and compare A - S
    T = oba.edge_terms(g["poses"], g["disps"], g["intrinsics"], g["targets"], g["weights"], ii, jj)
    Hs, vs, Eii, Eij, Cii, bz = oba.edge_blocks(T)
    P = N - 1
    HW = Cii.shape[1]
    Ed = np.zeros((P, 6, len(kx), HW))
    kk = np.searchsorted(kx, ii)
    for e in range(len(ii)):
        if ii[e] >= 1:
            Ed[ii[e] - 1, :, kk[e]] += Eii[e]
        if jj[e] >= 1:
            Ed[jj[e] - 1, :, kk[e]] += Eij[e]
    Q = 1.0 / info["C"]
    Ef = Ed.reshape(6 * P, -1)
    S = (Ef * Q.reshape(-1)) @ Ef.T
    A = info["A"].reshape(6 * P, 6 * P)
    assert np.abs((A - S) - info["H"]).max() < 1e-8 * max(1.0, np.abs(A).max())
    bS = Ef @ (Q.reshape(-1) * info["w"].reshape(-1))
    assert np.abs((info["bA"].reshape(-1) - bS) - info["b"]).max() < 1e-9 * max(1.0, np.abs(bS).max())


def test_corr_volume_pyramid_matches_reference_python(golden_dir):
    g = _load(golden_dir, "corr_python.npz")
    pyr = ocorr.corr_pyramid(g["fmap1"], g["fmap2"], 4)
    for l in range(4):
        ref = g["level%d" % l]
        assert pyr[l].shape == ref.shape
        assert np.abs(pyr[l] - ref).max() < 2e-5 * np.abs(ref).max()


def test_corr_lookup_equals_grid_sample(golden_dir):
    """corr_index_forward == F.grid_sample(align_corners=True, zeros) on each pixel's slice."""
    g = _load(golden_dir, "corr_python.npz")
    vol = g["level0"]
    N, h1, w1, h2, w2 = vol.shape
    rng = np.random.default_rng(3)
    coords = np.stack([rng.uniform(-3, w2 + 2, (N, h1, w1)), rng.uniform(-3, h2 + 2, (N, h1, w1))], 1).astype(np.float32)
    out = ocorr.corr_index_forward(vol, coords, 3)
    r = 3
    dxs = torch.arange(-r, r + 1, dtype=torch.float64)
    v = torch.as_tensor(vol, dtype=torch.float64).reshape(N * h1 * w1, 1, h2, w2)
    c = torch.as_tensor(coords, dtype=torch.float64).permute(0, 2, 3, 1).reshape(-1, 1, 1, 2)
    gx = c[..., 0] + dxs.view(1, -1, 1)          # [M, 7(a), 1]
    gy = c[..., 1] + dxs.view(1, 1, -1)          # [M, 1, 7(b)]
    gx, gy = torch.broadcast_tensors(gx, gy)
    grid = torch.stack([2 * gx / (w2 - 1) - 1, 2 * gy / (h2 - 1) - 1], -1)
    ref = torch.nn.functional.grid_sample(v, grid, align_corners=True, padding_mode="zeros")  # [M,1,7,7]
    ref = ref.reshape(N, h1, w1, 7, 7).permute(0, 3, 4, 1, 2).numpy()
    assert np.abs(out - ref).max() < 1e-6   # kernel semantics: fractional offsets formed in fp32


def test_altcorr_equals_lookup_on_materialised_volume():
    rng = np.random.default_rng(4)
    N, C, H, W = 3, 8, 8, 16
    fm = rng.standard_normal((1, N, C, H, W))
    ii = np.array([0, 1, 2, 2]); jj = np.array([1, 0, 0, 2])
    M = len(ii)
    coords = np.stack([rng.uniform(-2, W + 1, (1, M, H, W)), rng.uniform(-2, H + 1, (1, M, H, W))], -1).astype(np.float32)
    alt = ocorr.alt_block_lookup(fm, coords, ii, jj, radius=3, num_levels=3)
    # volume route: pooling the VOLUME over (y2,x2) == correlating with pooled features (linearity)
    pyr = ocorr.corr_pyramid(fm[0, ii], fm[0, jj], 3)
    vol = ocorr.corr_block_lookup(pyr, coords[0], radius=3)
    assert np.abs(alt[0] - vol).max() < 1e-10


def test_update_module_matches_reference_python(golden_dir):
    g = _load(golden_dir, "update_python.npz")
    sd = deterministic_state_dict(_Named(oupd.empty_state_dict()), seed=int(g["seed"]))
    t = lambda k: torch.as_tensor(g[k])
    with torch.no_grad():
        net1, delta, weight, eta, upmask = oupd.update_forward(sd, t("net"), t("inp"), t("corr"), t("flow"), t("ii"))
        up = oupd.cvx_upsample(t("disp"), upmask)
    assert (net1 - t("net1")).abs().max() < 2e-5
    assert (delta - t("delta")).abs().max() < 2e-5
    assert (weight - t("weight")).abs().max() < 2e-5
    assert (eta - t("eta")).abs().max() < 1e-6
    assert (upmask - t("upmask").float()).abs().max() < 2e-3       # golden stored as fp16
    assert (up - t("disp_up")).abs().max() < 2e-3


class _Named:
    """Minimal state_dict carrier so weights.deterministic_state_dict can be used without an nn.Module."""

    def __init__(self, sd):
        self._sd = sd

    def state_dict(self):
        return self._sd


def test_update_autocast_oracle_equals_reference_module(golden_dir):
    """oracle.update.update_forward(autocast=True) == the reference's UpdateModule under torch.autocast(fp16)
    (tests/golden/update_autocast_python.npz): same rounding points, so equal to the last fp16 bit."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_inputs import update_autocast_inputs, UPDATE_AUTOCAST
    G = _load(golden_dir, "update_autocast_python.npz")

    class _SD:
        def state_dict(self):
            return oupd.empty_state_dict()
    sd = deterministic_state_dict(_SD(), seed=UPDATE_AUTOCAST["weight_seed"])
    net, inp, corr, flow, ii, jj = update_autocast_inputs()
    with torch.no_grad():
        out = oupd.update_forward(sd, net, inp, corr, flow, ii, autocast=True)
    for name, t in zip(("net1", "delta", "weight", "eta", "upmask"), out):
        ref = G[name].astype(np.float32)
        assert np.abs(t.float().numpy() - ref).max() <= 2.0 ** -11 * max(1e-6, np.abs(ref).max()), name


def test_threaded_forms_of_the_oracle_equal_the_plain_ones():
    """bench.py's multi-core cpu_baseline uses oracle.ba(threads=..) and corr_block_lookup_torch: same numbers"""
    from droid_amd import synthetic as syn
    g = syn.small_graph(n_frames=6, seed=21, ht=12, wd=16)
    outs = []
    for kw in (dict(), dict(chunk=5), dict(threads=3)):
        p = g["poses"].astype(np.float64); d = np.array(g["disps"], dtype=np.float64, order="C")
        dx, dz = oba.ba(p, d, g["intrinsics"], g["disps_sens"], g["targets"], g["weights"], g["eta"], g["ii"], g["jj"], 1, 6, 2, 1e-4, 0.1, False, **kw)
        outs.append((p, d, dx, dz))
    for o in outs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], o))
    rng = np.random.default_rng(2)
    f1 = rng.standard_normal((2, 16, 8, 8)).astype(np.float32); f2 = rng.standard_normal((2, 16, 8, 8)).astype(np.float32)
    pyr = ocorr.corr_pyramid(f1, f2, 3)
    coords = np.stack([rng.uniform(-3, 10, (2, 8, 8)), rng.uniform(-3, 10, (2, 8, 8))], -1).astype(np.float32)
    a = ocorr.corr_block_lookup(pyr, coords, 3)
    b = ocorr.corr_block_lookup_torch([torch.as_tensor(v, dtype=torch.float32) for v in pyr], torch.as_tensor(coords), 3).numpy()
    assert np.abs(a - b).max() < 1e-4 * np.abs(a).max()


def test_encoder_oracle_equals_reference_module(golden_dir):
    """oracle.encoder.basic_encoder(autocast=True) == the reference's BasicEncoder under autocast (fnet 'instance', cnet 'none')"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_inputs import encoder_inputs
    from oracle import encoder as oenc
    from droid_amd.encoder import empty_state_dict
    G = _load(golden_dir, "encoder_python.npz")
    x = encoder_inputs()
    for tag, dim, inst, seed in (("fnet", 128, True, 4321), ("cnet", 256, False, 8765)):
        class _SD:
            def state_dict(self):
                return empty_state_dict(dim)
        sd = deterministic_state_dict(_SD(), seed=seed)
        with torch.no_grad():
            y = oenc.basic_encoder(sd, x, inst, autocast=True)
        ref = G[tag].astype(np.float32)
        assert np.abs(y.float().numpy() - ref).max() <= 2.0 ** -10 * np.abs(ref).max(), tag


def test_fast_altcorr_equals_altcorr():
    """oracle.corr.altcorr_forward_fast (GEMM + integer-tap gather; used only to make the C5-size golden in reasonable time) against
    altcorr_forward (the restatement pinned to the reference's kernel by ref_cuda.npz): coordinates leaving the image on every side,
    a pooled target level, stereo-style equal indices"""
    from oracle import corr as ocorr
    rng = np.random.default_rng(3)
    N, C, H, W = 3, 16, 8, 12
    f1 = rng.standard_normal((1, N, C, H, W)).astype(np.float32)
    for H2, W2 in ((H, W), (H // 2, W // 2)):
        f2 = rng.standard_normal((1, N, C, H2, W2)).astype(np.float32)
        ii = np.array([0, 1, 2, 1]); jj = np.array([1, 1, 0, 2])
        coords = np.stack([rng.uniform(-4, W2 + 4, (1, 4, H, W)), rng.uniform(-4, H2 + 4, (1, 4, H, W))], 2).astype(np.float32)
        a = ocorr.altcorr_forward(f1, f2, coords, ii, jj, 3)
        b = ocorr.altcorr_forward_fast(f1, f2, coords, ii, jj, 3, edge_chunk=3)
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(a).max())
