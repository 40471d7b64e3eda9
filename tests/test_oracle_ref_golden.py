"""Pin the CPU oracle against vectors produced by the REFERENCE's own CUDA kernels (tests/golden/ref_cuda.npz, written
on an MI355X by tests/golden/make_ref_golden.py from oracle/_ref = the reference's droid_kernels.cu /
correlation_kernels.cu / altcorr_kernel.cu compiled for gfx950 where they lie).  This is what pins the CUDA-only
semantics: damping on the reduced system, MIN_DEPTH 0.25, the EvT6x1 row skip, stereo edges, the alpha = 0.05 sensor
prior, "failure -> dx = 0", the fp16 lookup, the alt correlation's (unscaled) backward and the geometry kernels.

The reference computes in fp32 with its own reduction order, the oracle in fp64: tolerances are fp32 round-off of sums
over HW = 192 pixels, stated per quantity."""
import os
import numpy as np
import pytest

from oracle import ba as oba, corr as ocorr, geom as ogeom

CASES = ["mono", "stereo", "sensor", "t0_3", "global", "motion"]


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_cuda.npz"))


def _case(G, name):
    g = {k: G["ba_%s_%s" % (name, k)] for k in ("poses", "disps", "intrinsics", "disps_sens", "targets", "weights", "ii", "jj", "eta")}
    t0, t1, lm, ep, mo = G["ba_%s_args" % name]
    return g, int(t0), int(t1), float(lm), float(ep), bool(mo)


def _rot_angle(q, qr):
    v = q[:, 3:4] * -qr[:, :3] + qr[:, 3:4] * q[:, :3] + np.cross(q[:, :3], -qr[:, :3])
    return 2 * np.linalg.norm(v, axis=-1)


@pytest.mark.parametrize("name", CASES)
def test_per_edge_blocks_match_reference_kernel(G, name):
    """projective_transform_kernel (droid_kernels.cu:185-433): Hs, vs, Eii, Eij, Cii, bz, rel 1e-5 of each block's scale."""
    g, t0, t1, lm, ep, mo = _case(G, name)
    T = oba.edge_terms(g["poses"], g["disps"], g["intrinsics"], g["targets"], g["weights"], g["ii"], g["jj"])
    Hs, vs, Eii, Eij, Cii, bz = oba.edge_blocks(T)
    for nm, mine in (("Hs", Hs), ("vs", vs), ("Eii", Eii), ("Eij", Eij), ("Cii", Cii), ("bz", bz)):
        ref = G["ba_%s_%s" % (name, nm)].astype(np.float64)
        assert mine.shape == ref.shape, nm
        assert np.abs(mine - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-12, nm


@pytest.mark.parametrize("name", CASES)
def test_reduced_camera_system_matches_reference(G, name):
    """[A - S | b] right before SparseBlock::solve, and the depth blocks C, w (droid_kernels.cu:1385-1415)."""
    g, t0, t1, lm, ep, mo = _case(G, name)
    p = g["poses"].astype(np.float64); d = g["disps"].astype(np.float64)
    dx, dz, info = oba.ba(p, d, g["intrinsics"], g["disps_sens"], g["targets"], g["weights"], g["eta"], g["ii"], g["jj"],
                          t0, t1, 1, lm, ep, mo, return_system=True)
    if mo:
        pytest.skip("motion-only: the system is A itself, covered by the edge blocks")
    H, b = G["ba_%s_H" % name], G["ba_%s_b" % name]
    assert np.abs(info["H"] - H).max() <= 2e-5 * np.abs(H).max()
    assert np.abs(info["b"] - b).max() <= 2e-5 * np.abs(b).max() + 1e-8
    assert np.abs(info["C"] - G["ba_%s_C" % name]).max() <= 1e-5 * np.abs(info["C"]).max()
    assert np.abs(info["w"] - G["ba_%s_w" % name]).max() <= 1e-5 * np.abs(info["w"]).max() + 1e-9


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name", CASES)
def test_ba_iterations_match_reference(G, name, dtype):
    """ba_cuda (droid_kernels.cu:1323-1443), two Gauss-Newton iterations, each from the reference's own state
    (iteration 2 starts from the reference's result of iteration 1)."""
    g, t0, t1, lm, ep, mo = _case(G, name)
    state = (g["poses"], g["disps"])
    for it in (1, 2):
        p = state[0].astype(np.float64); d = state[1].astype(np.float64)
        dx, dz = oba.ba(p, d, g["intrinsics"], g["disps_sens"], g["targets"], g["weights"], g["eta"], g["ii"], g["jj"],
                        t0, t1, 1, lm, ep, mo, dtype=dtype)
        rp, rd, rdx = G["ba_%s_poses%d" % (name, it)], G["ba_%s_disps%d" % (name, it)], G["ba_%s_dx%d" % (name, it)]
        if it == 2:
            # the reference's dx of a 2-iteration call is the second iteration's
            pass
        assert np.linalg.norm(dx - rdx) <= 1e-3 * np.linalg.norm(rdx) + 1e-7
        assert np.abs(p[:, :3] - rp[:, :3]).max() <= 1e-4
        assert _rot_angle(p[:, 3:], rp[:, 3:].astype(np.float64)).max() <= 1e-4
        assert np.array_equal(p[:t0].astype(np.float32), g["poses"][:t0])
        if not mo:
            rdz = G["ba_%s_dz%d" % (name, it)]
            e = np.abs(dz - rdz) / np.maximum(1.0, np.abs(rdz))
            assert np.quantile(e, 0.995) <= 1e-4 and e.max() <= 1e-2
            e = np.abs(d - rd) / np.maximum(1.0, np.abs(rd))
            assert np.quantile(e, 0.995) <= 1e-4 and e.max() <= 1e-2
        else:
            assert np.array_equal(rd, g["disps"])
        state = (rp, rd)


def test_row_skip_quirk_is_what_the_reference_does(G):
    """EvT6x1_kernel skips rows whose relative pose index is <= 0 (droid_kernels.cu:1114): with the skip the oracle's dz
    matches the reference, without it it does not."""
    g, t0, t1, lm, ep, mo = _case(G, "mono")
    out = {}
    for strict in (True, False):
        p = g["poses"].astype(np.float64); d = g["disps"].astype(np.float64)
        _, dz = oba.ba(p, d, g["intrinsics"], g["disps_sens"], g["targets"], g["weights"], g["eta"], g["ii"], g["jj"],
                       t0, t1, 1, lm, ep, mo, strict_q6=strict)
        out[strict] = np.abs(dz - G["ba_mono_dz1"]).max()
    assert out[True] < 1e-3 * np.abs(G["ba_mono_dz1"]).max() < out[False]


def test_cholesky_failure_gives_zero_update_in_the_reference(G):
    assert np.all(G["fail_dx"] == 0) and bool(G["fail_poses_unchanged"])


def test_corr_index_forward_and_backward_match_reference_kernels(G):
    vol = G["ci_vol"].astype(np.float32)
    ref32, ref16 = G["ci_out_f32"], G["ci_out_f16"].astype(np.float32)
    mine = ocorr.corr_index_forward(vol, G["ci_coords"], 3)
    assert np.abs(mine - ref32).max() <= 2e-5
    # the reference accumulates the four bilinear contributions in fp16 in global memory (correlation_kernels.cu:56-66)
    assert np.abs(mine - ref16).max() <= 2.0 ** -8 * np.abs(ref32).max()
    vg = ocorr.corr_index_backward(vol.shape, G["ci_coords"], G["ci_grad"], 3)
    assert np.abs(vg - G["ci_vgrad"]).max() <= 1e-5


def test_altcorr_forward_and_backward_match_reference_kernels(G):
    fm = G["alt_fmap"].astype(np.float32)
    ii, jj = G["alt_ii"], G["alt_jj"]
    for lvl, f2 in ((0, fm), (1, G["alt_fmap_l1"].astype(np.float32))):
        c = G["alt_coords_l%d" % lvl]
        mine = ocorr.altcorr_forward(fm, f2, c, ii, jj, 3)
        ref32 = G["alt_out_f32_l%d" % lvl]
        assert mine.shape == ref32.shape
        assert np.abs(mine - ref32).max() <= 1e-5 * max(1.0, np.abs(ref32).max())
        ref16 = G["alt_out_f16_l%d" % lvl].astype(np.float32)
        assert np.abs(mine - ref16).max() <= 2.0 ** -8 * np.abs(ref32).max()
    g1, g2 = ocorr.altcorr_backward(fm, fm, G["alt_coords_l0"], G["alt_grad"], ii, jj, 3)
    assert np.abs(g1 - G["alt_g1"]).max() <= 1e-4 * np.abs(G["alt_g1"]).max()
    assert np.abs(g2 - G["alt_g2"]).max() <= 1e-4 * np.abs(G["alt_g2"]).max()


def test_geometry_kernels_match_reference(G):
    poses, disps, intr, ii, jj = G["geo_poses"], G["geo_disps"], G["geo_intr"], G["geo_ii"], G["geo_jj"]
    d = ogeom.frame_distance(poses, disps, intr, ii, jj, 0.3)
    assert np.abs(d - G["geo_dist"]).max() <= 1e-4 * max(1.0, np.abs(G["geo_dist"]).max())
    c, v = ogeom.projmap(poses, disps, intr, ii, jj)
    assert np.abs(c - G["geo_pm_coords"]).max() < 1e-3 and np.array_equal(v.astype(np.float32), G["geo_pm_valid"])
    pts = ogeom.iproj(poses, disps, intr)
    assert np.abs(pts - G["geo_points"]).max() <= 1e-5 * np.abs(G["geo_points"]).max()
    cnt = ogeom.depth_filter(poses, disps, intr, G["geo_ix"], G["geo_th"])
    assert np.mean(cnt != G["geo_count"]) < 5e-3          # threshold comparisons may flip on fp32 rounding
