"""GPU parity tests: the HIP path (through droid_backends -> C ABI) against the CPU oracle on identical
seeded inputs, and against the committed reference-Python golden vectors.  Run on the MI355X box:

    python -m pytest tests -m gpu -x -q

Tolerances: corr lookup fp32 <= 1e-5 abs (scaled), fp16 <= 2^-9 * max|corr|;
BA: dx  ||d||/||dx|| <= 1e-3, dz / disps rel 1e-3, poses |dt| <= 1e-4, rotation <= 1e-4 rad.
"""
import ctypes
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ba as oba, corr as ocorr, geom as ogeom, se3 as ose3
from droid_amd import synthetic as syn


@pytest.fixture(scope="module")
def db():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import droid_backends
    return droid_backends


@pytest.fixture
def option(db):
    """set library switches for one test (droid_backends.set_option), restored afterwards"""
    saved = {}

    def set_(name, value):
        saved.setdefault(name, db.get_option(name))
        db.set_option(name, int(value))
    yield set_
    for k, v in saved.items():
        db.set_option(k, v)


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


# ------------------------------------------------------------------------------------------ corr lookup
def _rand_coords(rng, N, h1, w1, h2, w2, spread=4.0):
    """mostly in-range, some far out of range, some exactly integral"""
    x = rng.uniform(-spread, w2 - 1 + spread, (N, h1, w1))
    y = rng.uniform(-spread, h2 - 1 + spread, (N, h1, w1))
    x[:, 0, 0] = 5.0; y[:, 0, 0] = 2.0
    x[:, -1, -1] = -50.0
    y[:, -1, 0] = 1e4
    return np.stack([x, y], 1).astype(np.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(3, 12, 16, 48, 64), (2, 6, 8, 24, 32), (2, 5, 7, 12, 16), (2, 4, 4, 6, 8),
                                   (2, 3, 5, 5, 7), (1, 2, 2, 2, 2)])
def test_corr_index_forward(db, dtype, shape):
    N, h1, w1, h2, w2 = shape
    rng = np.random.default_rng(hash(shape) % 1000)
    vol = rng.standard_normal(shape).astype(np.float32)
    coords = _rand_coords(rng, N, h1, w1, h2, w2)
    v = dev(vol, dtype)
    out, = db.corr_index_forward(v, dev(coords), 3)
    assert out.shape == (N, 7, 7, h1, w1) and out.dtype == dtype
    ref = ocorr.corr_index_forward(v.float().cpu().numpy(), coords, 3)
    tol = 2e-5 if dtype == torch.float32 else 2.0 ** -9 * np.abs(ref).max()
    assert np.abs(out.float().cpu().numpy() - ref).max() <= tol


@pytest.mark.parametrize("radius", [1, 2, 4])
def test_corr_index_forward_other_radius(db, radius):
    rng = np.random.default_rng(radius)
    shape = (2, 6, 8, 12, 16)
    vol = rng.standard_normal(shape).astype(np.float32)
    coords = _rand_coords(rng, *shape)
    out, = db.corr_index_forward(dev(vol), dev(coords), radius)
    ref = ocorr.corr_index_forward(vol, coords, radius)
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-5


def test_corr_index_empty_and_noncontiguous(db):
    out, = db.corr_index_forward(torch.zeros(0, 4, 4, 4, 4).cuda(), torch.zeros(0, 2, 4, 4).cuda(), 3)
    assert out.shape == (0, 7, 7, 4, 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        db.corr_index_forward(torch.zeros(2, 4, 4, 4, 8).cuda()[..., ::2], torch.zeros(2, 2, 4, 4).cuda(), 3)


def test_corr_index_backward_is_the_adjoint(db):
    rng = np.random.default_rng(5)
    shape = (2, 6, 8, 12, 16)
    coords = _rand_coords(rng, *shape, spread=2.0)
    g = rng.standard_normal((2, 7, 7, 6, 8)).astype(np.float32)
    vg, = db.corr_index_backward(torch.zeros(shape).cuda(), dev(coords), dev(g), 3)
    ref = ocorr.corr_index_backward(shape, coords, g, 3)
    assert np.abs(vg.cpu().numpy() - ref).max() < 1e-5
    # <lookup(v), g> == <v, adjoint(g)>
    vol = rng.standard_normal(shape).astype(np.float32)
    out, = db.corr_index_forward(dev(vol), dev(coords), 3)
    lhs = float((out.double().cpu() * torch.as_tensor(g).double()).sum())
    rhs = float((torch.as_tensor(vol).double() * vg.double().cpu()).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_full_size_lookup_properties(db):
    """BASELINE full size (48x64, one level-0 volume of 16 edges): linearity and the constant-volume identity."""
    torch.manual_seed(0)
    N, h, w = 16, 48, 64
    vol = torch.randn(N, h, w, h, w, device="cuda", dtype=torch.float16)
    coords = torch.stack([torch.rand(N, h, w, device="cuda") * (w + 6) - 3,
                          torch.rand(N, h, w, device="cuda") * (h + 6) - 3], 1).contiguous()
    a, = db.corr_index_forward(vol, coords, 3)
    b, = db.corr_index_forward(vol * 2, coords, 3)
    assert (b.float() - 2 * a.float()).abs().max() <= 2.0 ** -8 * a.float().abs().max()
    ones, = db.corr_index_forward(torch.ones_like(vol), coords, 3)
    inside = ((coords[:, 0] >= 3) & (coords[:, 0] <= w - 5) & (coords[:, 1] >= 3) & (coords[:, 1] <= h - 5))
    sel = ones.float().permute(0, 3, 4, 1, 2)[inside]
    assert (sel - 1).abs().max() < 2e-3


# ------------------------------------------------------------------------------------------ alt corr
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_altcorr_forward(db, dtype):
    rng = np.random.default_rng(8)
    B, N, C, H, W = 1, 4, 32, 12, 16
    fm = rng.standard_normal((B, N, C, H, W)).astype(np.float32)
    ii = np.array([0, 1, 3, 2, 2]); jj = np.array([1, 0, 3, 0, 3])
    M = len(ii)
    f1 = dev(fm, dtype)
    for lvl, (H2, W2) in enumerate([(12, 16), (6, 8)]):
        f2np = fm if lvl == 0 else ocorr.avg_pool2(fm.astype(np.float64)).astype(np.float32)
        f2 = dev(f2np, dtype)
        coords = np.stack([rng.uniform(-3, W2 + 2, (B, M, H, W)), rng.uniform(-3, H2 + 2, (B, M, H, W))], 2).astype(np.float32)
        out, = db.altcorr_forward(f1, f2, dev(coords), dev(ii), dev(jj), 3)
        assert out.shape == (B, M, 7, 7, H, W)
        ref = ocorr.altcorr_forward(f1.float().cpu().numpy(), f2.float().cpu().numpy(), coords, ii, jj, 3)
        tol = 1e-4 if dtype == torch.float32 else 2.0 ** -9 * np.abs(ref).max()
        assert np.abs(out.float().cpu().numpy() - ref).max() <= tol


def test_altcorr_matches_volume_lookup_at_full_size(db):
    """alt path == lookup into the materialised volume (48x64, C=128): the size-independent identity."""
    torch.manual_seed(1)
    N, C, H, W = 3, 128, 48, 64
    fm = torch.randn(1, N, C, H, W, device="cuda").half()
    ii = torch.tensor([0, 1, 2], device="cuda"); jj = torch.tensor([1, 2, 0], device="cuda")
    coords = torch.stack([torch.rand(1, 3, H, W, device="cuda") * (W + 4) - 2,
                          torch.rand(1, 3, H, W, device="cuda") * (H + 4) - 2], 2).contiguous()
    alt, = db.altcorr_forward(fm, fm, coords, ii, jj, 3)
    f1 = fm[0, ii].float().reshape(3, C, H * W) / 4
    f2 = fm[0, jj].float().reshape(3, C, H * W) / 4
    vol = torch.matmul(f1.transpose(1, 2), f2).reshape(3, H, W, H, W).contiguous()
    ref, = db.corr_index_forward(vol, coords[0].contiguous(), 3)
    assert (alt[0].float() - ref).abs().max() <= 2.0 ** -9 * ref.abs().max()


# ------------------------------------------------------------------------------------------ BA
def _run_ba(db, g, t0, t1, itrs, lm, ep, motion_only=False, eta=None):
    poses = dev(g["poses"]); disps = dev(g["disps"])
    eta_t = dev(g["eta"] if eta is None else eta)
    dx, dz = db.ba(poses, disps, dev(g["intrinsics"]), dev(g["disps_sens"]), dev(g["targets"]), dev(g["weights"]),
                   eta_t, dev(g["ii"]), dev(g["jj"]), t0, t1, itrs, lm, ep, motion_only)
    torch.cuda.synchronize()
    return poses.cpu().numpy(), disps.cpu().numpy(), dx.cpu().numpy(), dz.cpu().numpy()


def _oracle_ba(g, t0, t1, itrs, lm, ep, motion_only=False, eta=None):
    poses = g["poses"].astype(np.float64).copy(); disps = g["disps"].astype(np.float64).copy()
    dx, dz = oba.ba(poses, disps, g["intrinsics"], g["disps_sens"], g["targets"], g["weights"],
                    g["eta"] if eta is None else eta, g["ii"], g["jj"], t0, t1, itrs, lm, ep, motion_only)
    return poses, disps, dx, dz


def _eta_for(g, t0, t1):
    kx = np.unique(np.concatenate([np.arange(t0, t1), g["ii"]]))
    rng = np.random.default_rng(99)
    return (0.2 * rng.uniform(1e-6, 1e-3, (len(kx),) + g["disps"].shape[1:]) + 1e-7).astype(np.float32)


def _rot_angle(q, qr):
    """angle of q * conj(qr) from its vector part (well conditioned near identity, unlike arccos of the dot)."""
    v = q[:, 3:4] * -qr[:, :3] + qr[:, 3:4] * q[:, :3] + np.cross(q[:, :3], -qr[:, :3])
    return 2 * np.linalg.norm(v, axis=-1)


def _check_ba(got, ref, motion_only=False, scale=1.0):
    p, d, dx, dz = got
    rp, rd, rdx, rdz = ref
    assert np.linalg.norm(dx - rdx) <= 1e-3 * np.linalg.norm(rdx) + 1e-7
    assert np.abs(p[:, :3] - rp[:, :3]).max() <= 1e-4 * scale
    assert _rot_angle(p[:, 3:].astype(np.float64), rp[:, 3:]).max() <= 1e-4
    assert np.abs(np.linalg.norm(p[:, 3:], axis=-1) - 1).max() < 1e-4
    if not motion_only:
        assert np.abs(dz - rdz).max() <= 1e-2 * np.abs(rdz).max() + 1e-6
        assert np.quantile(np.abs(dz - rdz) / np.maximum(1.0, np.abs(rdz)), 0.995) <= 1e-4
        # near-epipole pixels (Jz -> 0 by cancellation) get huge, ill-conditioned depth steps r/Jz whose fp32
        # evaluation is only good to ~1e-3 relative; everything else must agree tightly
        err = np.abs(d - rd) / np.maximum(1.0, np.abs(rd))
        assert np.quantile(err, 0.995) <= 1e-4
        assert err.max() <= 1e-2


def _compare_per_iteration(db, g, t0, t1, iters, lm, ep, motion_only=False, eta=None):
    """Gauss-Newton on these synthetic problems is chaotic across iterations (a depth that crosses the
    Z < 0.25 cut flips a weight to zero), so every iteration is compared from the SAME input state:
    GPU(k iterations) vs oracle(1 iteration) started from GPU(k-1 iterations)."""
    state = dict(g)
    for k in range(1, iters + 1):
        got = _run_ba(db, g, t0, t1, k, lm, ep, motion_only=motion_only, eta=eta)
        ref = _oracle_ba(state, t0, t1, 1, lm, ep, motion_only=motion_only, eta=eta)
        _check_ba(got, ref, motion_only=motion_only)
        state = dict(g); state["poses"] = got[0]; state["disps"] = got[1]
    return got


@pytest.mark.parametrize("case", ["mono", "stereo", "sensor", "t0_3", "many_edges"])
def test_ba_small_graphs(db, case):
    kw = dict(n_frames=6, seed=21, ht=12, wd=16)
    t0 = 1
    if case == "stereo":
        kw.update(stereo=True)
    if case == "sensor":
        kw.update(sensor_depth=True)
    if case == "many_edges":
        kw.update(n_frames=14, radius=13)            # 13 out-edges per frame -> two Gram chunks
    g = syn.small_graph(**kw)
    N = g["n_frames"]
    if case == "t0_3":
        t0 = 3
    eta = _eta_for(g, t0, N)
    got = _compare_per_iteration(db, g, t0, N, 3, 1e-4, 0.1, eta=eta)
    # frames outside [t0,t1) keep their pose
    assert np.array_equal(got[0][:t0], g["poses"][:t0])


def test_ba_motion_only(db):
    g = syn.small_graph(n_frames=6, seed=4, ht=12, wd=16)
    got = _compare_per_iteration(db, g, 1, 6, 2, 1e-4, 0.1, motion_only=True)
    assert np.array_equal(got[1], g["disps"])            # depths untouched


def test_ba_config_c1(db):
    """BASELINE configs[0]: 8 keyframes / 32 edges at 48x64."""
    g = syn.make_graph("C1")
    _compare_per_iteration(db, g, 1, 8, 2, g["lm"], g["ep"])


def test_ba_config_c2_converges_and_matches(db):
    """BASELINE configs[1]: 64 keyframes / 512 edges.  One oracle iteration for parity, then the
    size-independent property: the weighted reprojection cost decreases under repeated BA."""
    g = syn.make_graph("C2")
    got = _run_ba(db, g, 1, 64, 1, g["lm"], g["ep"])
    ref = _oracle_ba(g, 1, 64, 1, g["lm"], g["ep"])
    _check_ba(got, ref)
    poses = dev(g["poses"]); disps = dev(g["disps"])
    args = [dev(g[k]) for k in ("intrinsics", "disps_sens", "targets", "weights", "eta", "ii", "jj")]
    c0 = oba.reprojection_cost(g["poses"], g["disps"], g["intrinsics"], g["targets"], g["weights"], g["ii"], g["jj"])
    costs = [c0]
    for _ in range(4):
        db.ba(poses, disps, *args, 1, 64, 2, g["lm"], g["ep"], False)
        disps.clamp_(min=0.001)
        costs.append(oba.reprojection_cost(poses.cpu().numpy(), disps.cpu().numpy(), g["intrinsics"],
                                           g["targets"], g["weights"], g["ii"], g["jj"]))
    assert costs[1] < 0.5 * costs[0]
    assert costs[-1] <= costs[1] * 1.01


def test_ba_cholesky_schedules_agree(db, option):
    """the look-ahead schedule (one fused launch per block column, the default) and the two-launch schedule apply the
    same updates to every block in the same order: identical factors, identical BA result (6 block columns at C2)"""
    g = syn.make_graph("C2")
    option("chol_lookahead", 0)
    ref = _run_ba(db, g, 1, 64, 2, g["lm"], g["ep"])
    option("chol_lookahead", 1)
    got = _run_ba(db, g, 1, 64, 2, g["lm"], g["ep"])
    for a, b in zip(got, ref):
        assert np.allclose(a, b, rtol=0, atol=1e-7)
    if db.get_option("ablation_build"):
        # the dataflow schedule of the -DDH_ABLATION build (chol_lookahead = 2: one persistent launch, ready flags instead of kernel
        # boundaries) applies the same updates in the same order as well: identical results, also at 47 block columns (C3)
        for cfg, t1 in ((g, 64), (syn.make_graph("C3"), 512)):
            option("chol_lookahead", 1)
            a = _run_ba(db, cfg, 1, t1, 2, cfg["lm"], cfg["ep"])
            for mode in (2, 3):             # 3: grouped acquires, operand blocks by LDS-DMA
                option("chol_lookahead", mode)
                b = _run_ba(db, cfg, 1, t1, 2, cfg["lm"], cfg["ep"])
                for x, y in zip(a, b):
                    assert np.array_equal(x, y), mode


def test_ba_cholesky_failure_gives_zero_update(db):
    """SparseBlock::solve semantics (src/droid_kernels.cu:1211-1219): not SPD -> dx = 0."""
    g = syn.small_graph(n_frames=5, seed=2, ht=12, wd=16)
    p, d, dx, dz = _run_ba(db, g, 1, 5, 1, 0.0, -1e9, motion_only=True)
    assert np.all(dx == 0) and np.array_equal(p, g["poses"])


def test_ba_is_run_to_run_stable(db):
    g = syn.make_graph("C1")
    a = _run_ba(db, g, 1, 8, 2, g["lm"], g["ep"])
    b = _run_ba(db, g, 1, 8, 2, g["lm"], g["ep"])
    assert np.abs(a[0] - b[0]).max() < 1e-6 and np.abs(a[1] - b[1]).max() < 1e-5


# ------------------------------------------------------------------------------------------ geometry
def test_geometry_kernels(db):
    g = syn.small_graph(n_frames=8, seed=13, ht=12, wd=16)
    poses, disps, intr = dev(g["poses_gt"]), dev(g["disps_gt"]), dev(g["intrinsics"])
    ii, jj = g["ii"], g["jj"]
    d = db.frame_distance(poses, disps, intr, dev(ii), dev(jj), 0.3).cpu().numpy()
    ref = ogeom.frame_distance(g["poses_gt"], g["disps_gt"], g["intrinsics"], ii, jj, 0.3)
    assert np.abs(d - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    c, v = db.projmap(poses, disps, intr, dev(ii), dev(jj))
    rc, rv = ogeom.projmap(g["poses_gt"], g["disps_gt"], g["intrinsics"], ii, jj)
    assert np.abs(c.cpu().numpy() - rc).max() < 1e-3 and np.array_equal(v.cpu().numpy(), rv.astype(np.float32))
    pts = db.iproj(poses, disps, intr).cpu().numpy()
    rp = ogeom.iproj(g["poses_gt"], g["disps_gt"], g["intrinsics"])
    assert np.abs(pts - rp).max() <= 1e-4 * np.abs(rp).max()
    ix = np.arange(8); th = np.full(8, 0.05, dtype=np.float32)
    cnt = db.depth_filter(poses, disps, intr, dev(ix), dev(th)).cpu().numpy()
    rcnt = ogeom.depth_filter(g["poses_gt"], g["disps_gt"], g["intrinsics"], ix, th)
    assert np.mean(cnt != rcnt) < 0.01            # threshold comparisons may flip on fp32 rounding
    assert cnt.max() <= 6 and cnt.min() >= 0


def test_reproject_matches_reference_python_golden(db, golden_dir):
    g = np.load(os.path.join(golden_dir, "ba_python.npz"))
    coords, valid = db.reproject(dev(g["poses"], torch.float32), dev(g["disps"], torch.float32),
                                 dev(g["intrinsics"], torch.float32), dev(g["ii"]), dev(g["jj"]))
    assert np.abs(coords.cpu().numpy() - g["coords"]).max() < 2e-4
    assert np.mean(valid.cpu().numpy() != g["valid"]) < 1e-3


def test_se3_ops(db):
    rng = np.random.default_rng(6)
    n = 100
    a = ose3.random_se3(rng, n); b = ose3.random_se3(rng, n)
    xi = rng.normal(0, 0.3, (n, 6)); xi[0] = 0; xi[1, 3:] = 1e-6
    A, Bt, XI = dev(a, torch.float32), dev(b, torch.float32), dev(xi, torch.float32)
    t, q = ose3.se3_inv(a[:, :3], a[:, 3:])
    assert np.abs(db.se3_op("inv", A, A).cpu().numpy() - np.concatenate([t, q], -1)).max() < 1e-5
    t, q = ose3.se3_mul(a[:, :3], a[:, 3:], b[:, :3], b[:, 3:])
    assert np.abs(db.se3_op("mul", A, Bt).cpu().numpy() - np.concatenate([t, q], -1)).max() < 1e-5
    t, q = ose3.se3_exp(xi)
    assert np.abs(db.se3_op("exp", XI, XI).cpu().numpy() - np.concatenate([t, q], -1)).max() < 1e-5
    t, q = ose3.se3_retr(xi, a[:, :3], a[:, 3:])
    assert np.abs(db.se3_op("retr", XI, A).cpu().numpy() - np.concatenate([t, q], -1)).max() < 1e-5
    X = rng.normal(0, 1, (n, 7, 4)); J = rng.normal(0, 1, (n, 7, 6))
    Y = db.se3_map("act4", A, dev(X, torch.float32)).cpu().numpy()
    assert np.abs(Y - ose3.se3_act(a[:, None, :3], a[:, None, 3:], X)).max() < 1e-5
    Z = db.se3_map("adjT", A, dev(J, torch.float32)).cpu().numpy()
    assert np.abs(Z - ose3.se3_adjT(a[:, None, :3], a[:, None, 3:], J)).max() < 2e-5


# ------------------------------------------------------------------------------------------ raw C ABI
def test_raw_c_abi_call(db):
    """Call the C entry point directly (ctypes, raw device pointers, explicit stream)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = ctypes.CDLL(os.path.join(root, "droid-slam_amd", "libdroid_hip.so"))
    rng = np.random.default_rng(1)
    shape = (2, 6, 8, 12, 16)
    vol = rng.standard_normal(shape).astype(np.float32)
    coords = _rand_coords(rng, *shape)
    v, c = dev(vol), dev(coords)
    out = torch.empty(2, 7, 7, 6, 8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.dh_corr_index_fwd(ctypes.c_void_p(v.data_ptr()), ctypes.c_void_p(c.data_ptr()),
                               ctypes.c_void_p(out.data_ptr()), 1, 2, 6, 8, 12, 16, 3, ctypes.c_void_p(st))
    assert rc == 0
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - ocorr.corr_index_forward(vol, coords, 3)).max() < 2e-5


def _capi():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = ctypes.CDLL(os.path.join(root, "droid-slam_amd", "libdroid_hip.so"))
    lib.dh_ba_workspace_bytes.restype = ctypes.c_size_t
    lib.dh_corr_pyramid_bytes.restype = ctypes.c_size_t
    lib.dh_corr_pyramid_workspace_bytes.restype = ctypes.c_size_t
    return lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_raw_c_abi_ba_on_a_side_stream(db):
    """dh_ba (droid.cpp:93-122's replacement) through ctypes: raw device pointers, caller-allocated workspace of
    dh_ba_workspace_bytes, a NON-default stream -- equal to droid_backends.ba (the torch binding) bit for bit."""
    lib = _capi()
    g = syn.make_graph("C1")
    F, ht, wd = g["disps"].shape
    E, t0, t1 = len(g["ii"]), 1, g["n_frames"]
    K = g["eta"].shape[0]
    want_p, want_d = dev(g["poses"]), dev(g["disps"])
    wdx, wdz = db.ba(want_p, want_d, dev(g["intrinsics"]), dev(g["disps_sens"]), dev(g["targets"]), dev(g["weights"]), dev(g["eta"]),
                     dev(g["ii"]), dev(g["jj"]), t0, t1, 2, g["lm"], g["ep"], False)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    poses, disps = dev(g["poses"]), dev(g["disps"])
    intr, sens, tg, wt, eta, ii, jj = (dev(g[k]) for k in ("intrinsics", "disps_sens", "targets", "weights", "eta", "ii", "jj"))
    wsb = lib.dh_ba_workspace_bytes(F, E, ht, wd, t0, t1, 0)
    assert wsb > 0
    ws = torch.empty(wsb + 256, dtype=torch.uint8, device="cuda")
    off = (-ws.data_ptr()) % 256
    dx = torch.zeros(t1 - t0, 6, device="cuda"); dz = torch.zeros(K, ht * wd, device="cuda")
    torch.cuda.synchronize()
    # a mis-aligned or short workspace is refused before anything is launched
    assert lib.dh_ba(_p(poses), _p(disps), _p(intr), _p(sens), _p(tg), _p(wt), _p(eta), _p(ii), _p(jj), F, E, K, ht, wd, t0, t1, 2,
                     ctypes.c_float(g["lm"]), ctypes.c_float(g["ep"]), 0, _p(dx), _p(dz), ctypes.c_void_p(ws.data_ptr() + off), ctypes.c_size_t(wsb - 1),
                     ctypes.c_void_p(side.cuda_stream)) != 0
    rc = lib.dh_ba(_p(poses), _p(disps), _p(intr), _p(sens), _p(tg), _p(wt), _p(eta), _p(ii), _p(jj), F, E, K, ht, wd, t0, t1, 2,
                   ctypes.c_float(g["lm"]), ctypes.c_float(g["ep"]), 0, _p(dx), _p(dz), ctypes.c_void_p(ws.data_ptr() + off), ctypes.c_size_t(wsb),
                   ctypes.c_void_p(side.cuda_stream))
    assert rc == 0
    side.synchronize()
    assert torch.equal(poses, want_p) and torch.equal(disps, want_d) and torch.equal(dx, wdx) and torch.equal(dz, wdz)


def test_raw_c_abi_altcorr_and_fused_lookup_on_a_side_stream(db):
    """dh_altcorr_fwd (droid.cpp:198-204's replacement) and dh_corr_pyramid_build + dh_corr_pyramid_lookup_corr0 through
    ctypes on a non-default stream == the torch binding's results"""
    from droid_amd.update import pack_corr0_fused
    lib = _capi()
    rng = np.random.default_rng(9)
    side = torch.cuda.Stream()
    # ---- altcorr_forward: fmap1 [B,N,C,H,W], fmap2 [B,N,C,H2,W2], coords [B,M,2,H,W], ii, jj [M] -> [B,M,7,7,H,W]
    B, N, C, H, W, M = 1, 3, 128, 8, 16, 5
    f1 = dev(rng.standard_normal((B, N, C, H, W)).astype(np.float32)); f2 = dev(rng.standard_normal((B, N, C, H, W)).astype(np.float32))
    coords = dev(np.stack([rng.uniform(-2, W + 1, (B, M, H, W)), rng.uniform(-2, H + 1, (B, M, H, W))], 2).astype(np.float32))
    ii = dev(rng.integers(0, N, M)); jj = dev(rng.integers(0, N, M))
    want, = db.altcorr_forward(f1, f2, coords, ii, jj, 3)
    want = want.contiguous()
    torch.cuda.synchronize()
    out = torch.zeros(B, M, 7, 7, H, W, device="cuda")
    torch.cuda.synchronize()
    rc = lib.dh_altcorr_fwd(_p(f1), _p(f2), _p(coords), _p(ii), _p(jj), _p(out), 1, B, N, N, C, H, W, H, W, M, 3, ctypes.c_void_p(side.cuda_stream))
    assert rc == 0
    side.synchronize()
    # the binding returns the reference's permuted view [B,M,H,W,7,7] -> compare in the C ABI's own layout
    w = want if want.shape == out.shape else want.permute(0, 1, 4, 5, 2, 3).contiguous()
    assert torch.equal(out, w)
    # ---- pyramid build + the fused lookup
    E, h, wd = 3, 16, 64
    g1 = dev(rng.standard_normal((E, 128, h, wd)).astype(np.float16)); g2 = dev(rng.standard_normal((E, 128, h, wd)).astype(np.float16))
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(wd, dtype=np.float32), indexing="ij")
    cc = dev(np.stack([np.stack([xx + 2.3 * (e + 1) + 0.07 * yy, yy - 1.4 * e + 0.02 * xx], -1) for e in range(E)]).astype(np.float32))
    wgt = dev((0.05 * rng.standard_normal((128, 196))).astype(np.float32)); bias = dev((0.3 * rng.standard_normal(128)).astype(np.float32))
    wpk = pack_corr0_fused(wgt)
    want_pyr = db.corr_pyramid_build(g1, g2)
    want_c0 = db.corr_pyramid_lookup_corr0(want_pyr, cc, wpk, bias)
    torch.cuda.synchronize()
    nbytes = lib.dh_corr_pyramid_bytes(E, h, wd); wsb = lib.dh_corr_pyramid_workspace_bytes(E, h, wd)
    assert nbytes == want_pyr.numel() * 2
    pyr = torch.zeros(nbytes // 2, dtype=torch.float16, device="cuda"); wsp = torch.empty(max(wsb, 1), dtype=torch.uint8, device="cuda")
    c0 = torch.zeros(E, h, wd, 128, dtype=torch.float16, device="cuda")
    torch.cuda.synchronize()
    st = ctypes.c_void_p(side.cuda_stream)
    assert lib.dh_corr_pyramid_build(_p(g1), _p(g2), _p(pyr), _p(wsp), ctypes.c_size_t(wsb), E, 128, h, wd, st) == 0
    assert lib.dh_corr_pyramid_lookup_corr0(_p(pyr), _p(cc), _p(wpk), _p(bias), _p(c0), E, h, wd, st) == 0
    side.synchronize()
    assert torch.equal(pyr, want_pyr.reshape(-1)) and torch.equal(c0, want_c0)


# ------------------------------------------------------------------------------------------ sharded BA
def test_sharded_ba_equals_single_gpu(db):
    """Edge-sharded BA emulated on one GPU: per-shard ba_build, summed systems (what the RCCL
    all-reduce does), per-shard ba_finish with depth ownership == one fused ba call."""
    from droid_amd.dist_ba import shard_edges_by_source_frame, local_eta_rows
    g = syn.make_graph("C1")
    N, t0, t1 = 8, 1, 8
    ref_p, ref_d, ref_dx, _ = _run_ba(db, g, t0, t1, 1, g["lm"], g["ep"])
    world = 3
    shards, bounds = shard_edges_by_source_frame(g["ii"], world)
    intr, sens = dev(g["intrinsics"]), dev(g["disps_sens"])
    states = []
    for r in range(world):
        e = shards[r]
        rows, _ = local_eta_rows(g["ii"], g["ii"][e], t0, t1)
        poses, disps = dev(g["poses"]), dev(g["disps"])
        jj = dev(g["jj"][e])
        eta = dev(g["eta"][rows])
        ws, system = db.ba_build(poses, disps, intr, sens, dev(g["targets"][e]), dev(g["weights"][e]), eta,
                                 dev(g["ii"][e]), jj, t0, t1, False)
        states.append((poses, disps, jj, ws, system, eta.shape[0]))
    total = sum(s[4] for s in states)
    out_d = torch.as_tensor(g["disps"]).cuda()
    for r, (poses, disps, jj, ws, system, nrows) in enumerate(states):
        system.copy_(total)
        dx, dz = db.ba_finish(poses, disps, jj, ws, nrows, t0, t1, g["lm"], g["ep"], False)
        lo, hi = bounds[r], min(bounds[r + 1], N)
        out_d[lo:hi] = disps[lo:hi]
        assert np.linalg.norm(dx.cpu().numpy() - ref_dx) <= 1e-4 * np.linalg.norm(ref_dx)
        assert np.abs(poses.cpu().numpy() - ref_p).max() < 1e-5
    assert np.abs(out_d.cpu().numpy() - ref_d).max() <= 1e-4 * max(1.0, np.abs(ref_d).max())


def test_dist_ba_world1_equals_fused(db):
    from droid_amd.dist_ba import DistBA
    g = syn.make_graph("C1")
    ref_p, ref_d, ref_dx, ref_dz = _run_ba(db, g, 1, 8, 2, g["lm"], g["ep"])
    poses, disps = dev(g["poses"]), dev(g["disps"])
    dx, dz = DistBA(world=1).ba(poses, disps, dev(g["intrinsics"]), dev(g["disps_sens"]), dev(g["targets"]),
                                dev(g["weights"]), dev(g["eta"]), dev(g["ii"]), dev(g["jj"]), 1, 8, 2, g["lm"], g["ep"])
    assert np.abs(poses.cpu().numpy() - ref_p).max() < 1e-6
    assert np.abs(disps.cpu().numpy() - ref_d).max() < 1e-5


# ------------------------------------------------------------------------------------------ native pyramid
def _smooth_coords(rng, E, h, w, amp=6.0):
    """coherent flow field (what a reprojection produces) + a few outliers / out-of-image pixels"""
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    c = np.zeros((E, h, w, 2), dtype=np.float64)
    for e in range(E):
        a = rng.uniform(-amp, amp, 6)
        c[e, ..., 0] = xx + a[0] + a[1] * xx / w + a[2] * yy / h
        c[e, ..., 1] = yy + a[3] + a[4] * xx / w + a[5] * yy / h
    c[:, 0, 0] = [-30.0, 5.0]
    c[:, -1, -1] = [w + 2.5, h - 0.5]
    return c.astype(np.float32)


@pytest.mark.parametrize("shape", [(3, 16, 16), (2, 8, 32), (2, 24, 16), (2, 48, 64)])
@pytest.mark.parametrize("kind", ["smooth", "random"])
def test_native_corr_pyramid_vs_oracle(db, shape, kind):
    from droid_amd.corr import CorrBlock
    E, h, w = shape
    rng = np.random.default_rng(E * 1000 + h + w)
    f1 = rng.standard_normal((E, 128, h, w)).astype(np.float16)
    f2 = rng.standard_normal((E, 128, h, w)).astype(np.float16)
    if kind == "smooth":
        coords = _smooth_coords(rng, E, h, w)
    else:
        coords = np.stack([rng.uniform(-4, w + 3, (E, h, w)), rng.uniform(-4, h + 3, (E, h, w))], -1).astype(np.float32)
    blk = CorrBlock(dev(f1)[None], dev(f2)[None])
    out = blk(dev(coords)[None])[0].float().cpu().numpy()
    assert out.shape == (E, 196, h, w)
    pyr = ocorr.corr_pyramid(f1, f2, 4)
    ref = ocorr.corr_block_lookup(pyr, coords, 3)
    assert np.abs(out - ref).max() <= 2.0 ** -9 * np.abs(ref).max()


def test_native_pyramid_matches_reference_layout_path_full_size(db):
    """same features, same coords: fused lookup on the skewed layout == 4 corr_index_forward launches"""
    from droid_amd.corr import CorrBlock, CorrBlockRef
    torch.manual_seed(3)
    E, h, w = 6, 48, 64
    f1 = torch.randn(1, E, 128, h, w, device="cuda").half()
    f2 = torch.randn(1, E, 128, h, w, device="cuda").half()
    rng = np.random.default_rng(0)
    coords = dev(_smooth_coords(rng, E, h, w))[None]
    a = CorrBlock(f1, f2)(coords).float()
    b = CorrBlockRef(f1, f2)(coords).float()
    assert a.shape == b.shape == (1, E, 196, h, w)
    assert (a - b).abs().max() <= 2.0 ** -9 * b.abs().max()


def test_native_pyramid_build_kernels_are_bit_identical(db, option):
    """the row-ring build kernel (default) and the chunk kernel (option pyr_build_chunk) accumulate every cell in the same
    order: the stored pyramids must be equal bit for bit, at all three supported widths"""
    torch.manual_seed(11)
    for (E, h, w) in [(3, 48, 64), (2, 16, 32), (2, 8, 16), (1, 24, 64)]:
        f1 = torch.randn(E, 128, h, w, device="cuda").half()
        f2 = torch.randn(E, 128, h, w, device="cuda").half()
        b = db.corr_pyramid_build(f1, f2)
        if db.get_option("ablation_build"):              # the chunk kernel exists only in a -DDH_ABLATION build
            option("pyr_build_chunk", 1)
            a = db.corr_pyramid_build(f1, f2)
            option("pyr_build_chunk", 0)
            assert torch.equal(a, b)
        option("pyr_build_waves", 4)                     # w = 64: the four-wave form of the row-ring kernel (8 waves is the default)
        c = db.corr_pyramid_build(f1, f2)
        option("pyr_build_waves", 8)
        assert torch.equal(b, c)
        # round 6: level 0 with tile-major wave roles (the default, `b`) against the row-pair-major roles of rounds 2-5, 8 and 4 waves
        option("pyr_build_tm", 0)
        d8 = db.corr_pyramid_build(f1, f2)
        option("pyr_build_waves", 4)
        d4 = db.corr_pyramid_build(f1, f2)
        option("pyr_build_waves", 8); option("pyr_build_tm", 1)
        assert torch.equal(b, d8) and torch.equal(b, d4)


def test_pyramid_build_with_an_edges_workgroups_on_one_xcd_is_bit_identical(db, option):
    """round 6 (option pyr_build_xcd): the build's workgroups are re-numbered so that the source blocks of one edge run on ONE XCD
    (its target rows then come from one L2: fabric reads 11.1 -> 0.9 GB per 256 edges, profiles/r06_v_pyr_build_pmc.txt).  A pure
    re-numbering: equal records, also when the edge count is no multiple of 8 (the last E % 8 edges keep the plain order)."""
    torch.manual_seed(12)
    for (E, h, w) in [(19, 16, 32), (8, 8, 16), (9, 48, 64), (16, 24, 64)]:
        f1 = torch.randn(E, 128, h, w, device="cuda").half()
        f2 = torch.randn(E, 128, h, w, device="cuda").half()
        option("pyr_build_xcd", 0)
        a = db.corr_pyramid_build(f1, f2)
        option("pyr_build_xcd", 1)
        b = db.corr_pyramid_build(f1, f2)
        assert torch.equal(a, b), (E, h, w)
        # two horizontally adjacent source blocks per workgroup sharing each staged target row (option pyr_build_dual, w = 64 only)
        for xcd in (0, 1):
            option("pyr_build_xcd", xcd); option("pyr_build_dual", 1)
            c = db.corr_pyramid_build(f1, f2)
            option("pyr_build_dual", 0)
            assert torch.equal(a, c), (E, h, w, xcd)


@pytest.mark.parametrize("shape,rig", [((48, 64), 1), ((16, 32), 2), ((30, 40), 1), ((24, 16), 2), ((41, 73), 1), ((72, 96), 1)])
def test_pyramid_from_frames_is_bit_identical_to_the_per_edge_build(db, shape, rig):
    """round 5: CorrBlock.from_frames(video.fmaps, ii, jj) -- features transposed and pooled once per frame
    (dh_corr_pyramid_prepare_frames), the build kernel indexed by the edges' frames (dh_corr_pyramid_build_indexed) -- against
    CorrBlock(fmaps[ii, 0], fmaps[jj, c]) (factor_graph.py:128-133): the same records bit for bit, mono and stereo (c = 1 on the
    self-edges), on canvases, into an arena; transposed / strip image sizes fall back to the per-edge constructor."""
    from droid_amd.corr import CorrBlock
    h, w = shape
    torch.manual_seed(h * 7 + w + rig)
    N = 6
    fmaps = torch.randn(N, rig, 128, h, w, device="cuda").half()
    ii = torch.tensor([0, 1, 1, 2, 3, 3, 4, 5, 5, 2], device="cuda")
    jj = torch.tensor([1, 0, 2, 1, 3 if rig > 1 else 4, 5, 3, 4, 5 if rig > 1 else 0, 4], device="cuda")
    c = (ii == jj).long() if rig > 1 else torch.zeros_like(ii)
    want = CorrBlock(fmaps[ii, 0][None], fmaps[jj, c][None])
    got = CorrBlock.from_frames(fmaps, ii, jj)
    torch.cuda.synchronize()
    assert (got.transposed, got.strips, got.hc, got.wc) == (want.transposed, want.strips, want.hc, want.wc)
    if want.strips is not None:
        assert all(torch.equal(a, b) for a, b in zip(got.records, want.records))
    else:
        assert got.pyramid.shape == want.pyramid.shape and torch.equal(got.pyramid, want.pyramid)
        if not want.transposed:
            arena = CorrBlock.arena(len(ii) + 3, h, w, "cuda")
            again = CorrBlock.from_frames(fmaps, ii, jj, out=arena)
            torch.cuda.synchronize()
            assert again.pyramid.data_ptr() == arena.data_ptr() and torch.equal(again.pyramid, want.pyramid)
    yy, xx = torch.meshgrid(torch.arange(h, device="cuda", dtype=torch.float32), torch.arange(w, device="cuda", dtype=torch.float32), indexing="ij")
    coords = (torch.stack([xx, yy], -1)[None, None] + torch.tensor([1.3, -2.1], device="cuda")).expand(1, len(ii), h, w, 2).contiguous()
    assert torch.equal(got(coords), want(coords))
    with pytest.raises(RuntimeError):                         # a frame index outside the prepared tensor is refused, not read
        prep = db.corr_pyramid_prepare_frames(torch.zeros(2, 128, 16, 16, device="cuda", dtype=torch.float16))
        db.corr_pyramid_build_indexed(prep, torch.tensor([0, 2], device="cuda"), torch.tensor([1, 1], device="cuda"), 16, 16)


def test_native_pyramid_cat_and_index(db):
    from droid_amd.corr import CorrBlock
    torch.manual_seed(4)
    h, w = 16, 16
    f1 = torch.randn(1, 5, 128, h, w, device="cuda").half()
    f2 = torch.randn(1, 5, 128, h, w, device="cuda").half()
    coords = dev(_smooth_coords(np.random.default_rng(1), 5, h, w))[None]
    full = CorrBlock(f1, f2)(coords)
    a = CorrBlock(f1[:, :2], f2[:, :2]).cat(CorrBlock(f1[:, 2:], f2[:, 2:]))
    assert torch.equal(a(coords), full)
    mask = torch.tensor([True, False, True, True, False], device="cuda")
    sub = a[mask]
    assert torch.equal(sub(coords[:, mask]), full[:, mask])


# ------------------------------------------------------------------------------------------ ConvGRU update operator
class _SD:
    def __init__(self, sd):
        self._sd = sd

    def state_dict(self):
        return self._sd


def _update_inputs(E, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    net = torch.tanh(torch.randn(E, 128, h, w, generator=g))
    inp = torch.relu(torch.randn(E, 128, h, w, generator=g))
    corr = torch.randn(E, 196, h, w, generator=g) * 2.0
    flow = (torch.randn(E, 4, h, w, generator=g) * 4.0).clamp(-64, 64)
    return net.half().float(), inp.half().float(), corr.half().float(), flow.half().float()


def _check_update(out, ref, K):
    """HIP update operator vs a reference evaluated under fp16 autocast.  Both sides store every layer output in fp16
    (fp32 accumulation inside a convolution); the HIP path fuses the GRU algebra into the convolution epilogue in fp32
    and rounds once where autocast rounds after every elementwise op, so they differ by a few fp16 roundings of O(1)
    activations: 2^-9 absolute on the tanh/sigmoid-bounded outputs, 2^-9 relative to the tensor's scale elsewhere
    (SURVEY 8c asks 2^-9 for fp16 quantities)."""
    n, d, wt, eta, up = out
    rn, rd, rw, re, ru = [t.float() for t in ref]
    cmp = lambda a, b: (a.float().cpu() - b).abs().max().item()
    tol = 2.0 ** -9
    assert cmp(n, rn) <= tol
    assert cmp(d, rd) <= tol * max(1.0, rd.abs().max().item())
    assert cmp(wt, rw) <= tol
    assert cmp(eta, re) <= tol * re.abs().max().item() + 1e-6
    assert cmp(up, ru) <= tol * max(1.0, ru.abs().max().item())


@pytest.mark.parametrize("share", [False, True])
@pytest.mark.parametrize("shape", [(6, 16, 16), (5, 12, 16), (3, 48, 64)])
def test_update_operator_vs_oracle(db, shape, share):
    """UpdateModule (implicit-GEMM MFMA convolutions + fused GRU epilogues) vs the oracle restatement of
    droid_net.py:111-143 evaluated under fp16 autocast like the reference's caller (factor_graph.py:214), on identical
    weights and fp16 inputs; the autocast oracle is pinned bit-exactly to the reference's own module
    (tests/test_oracle_golden.py::test_update_autocast_oracle_equals_reference_module)."""
    from oracle import update as oupd
    from droid_amd.update import UpdateModule
    from droid_amd.weights import deterministic_state_dict
    E, h, w = shape
    sd = deterministic_state_dict(_SD(oupd.empty_state_dict()), seed=7)
    net, inp, corr, flow = _update_inputs(E, h, w, seed=E + h)
    ii = torch.tensor(([0, 0, 1, 2, 2, 2] * 3)[:E], dtype=torch.int64)
    if share:      # context features per source frame (what the reference's callers pass): one convolution per frame
        inp = inp[[int(torch.nonzero(ii == f)[0]) for f in ii.tolist()]]
    with torch.no_grad():
        ref = oupd.update_forward(sd, net.half(), inp.half(), corr.half(), flow, ii, autocast=True)
    mod = UpdateModule(share_inp_by_source_frame=share).load_state_dict(sd)
    n, d, wt, eta, up = mod(net[None].cuda().half(), inp[None].cuda().half(), corr[None].cuda().half(),
                            flow[None].cuda(), ii.cuda(), None)
    torch.cuda.synchronize()
    assert n.shape == (1, E, 128, h, w) and d.shape == (1, E, h, w, 2) and wt.shape == (1, E, h, w, 2)
    K = len(torch.unique(ii))
    assert eta.shape == (1, K, h, w) and up.shape == (1, K, 576, h, w)
    _check_update((n[0], d[0], wt[0], eta[0], up[0]), ref, K)


@pytest.mark.parametrize("share", [False, True])
def test_update_operator_vs_reference_module_under_autocast(db, golden_dir, share):
    """the same operator against vectors written by the REFERENCE's own UpdateModule under torch.autocast(fp16)
    (tests/golden/make_golden.py update_autocast; W = 64: the production convolution kernels)"""
    from oracle import update as oupd
    from droid_amd.update import UpdateModule
    from droid_amd.weights import deterministic_state_dict
    from golden_inputs import update_autocast_inputs, UPDATE_AUTOCAST
    G = np.load(os.path.join(golden_dir, "update_autocast_python.npz"))
    sd = deterministic_state_dict(_SD(oupd.empty_state_dict()), seed=UPDATE_AUTOCAST["weight_seed"])
    net, inp, corr, flow, ii, jj = update_autocast_inputs()
    mod = UpdateModule().load_state_dict(sd)
    if share:
        # the golden inputs carry DIFFERENT context features on the two edges of frame 0, so the frame-level path is fed
        # through its explicit interface: one "frame" per edge (exercises the accumulator start values, index = identity)
        E = net.shape[0]
        nh = lambda t: mod.to_nhwc(t.cuda())
        cpad = torch.cat([corr.cuda(), torch.zeros_like(corr[:, :1]).cuda()], 1)
        c = nh(cpad[:, torch.where(mod.cmap >= 0, mod.cmap, torch.full_like(mod.cmap, 196))])
        c = c.view(E, net.shape[2], net.shape[3], 4, 56).permute(3, 0, 1, 2, 4).contiguous()
        n, d, wt, eta, up = mod.forward_nhwc(nh(net), None, c, mod.to_nhwc(flow.cuda(), 8), ii.cuda(),
                                             inp_frames=nh(inp), inp_index=torch.arange(E, device="cuda"))
        n, up = n.permute(0, 3, 1, 2)[None], up.permute(0, 3, 1, 2)[None]
        d, wt, eta = d[None], wt[None], eta[None]
    else:
        n, d, wt, eta, up = mod(net[None].cuda(), inp[None].cuda(), corr[None].cuda(), flow[None].cuda(), ii.cuda(), jj.cuda())
    torch.cuda.synchronize()
    ref = [torch.as_tensor(G[k].astype(np.float32)) for k in ("net1", "delta", "weight", "eta", "upmask")]
    _check_update((n[0], d[0], wt[0], eta[0], up[0]), ref, 2)


@pytest.mark.parametrize("halo,dma,halo2", [("1", "0", "1"), ("1", "0", "0"), ("1", "1", "1"), ("0", "0", "1")])
def test_conv2d_nhwc_matches_torch_conv(db, option, halo, dma, halo2):
    # all four main loops: halo2 (DMA weights, the default for 128-cout tiles), halo-tile, opt-in LDS-DMA, generic (fallback)
    if dma == "1" and not db.get_option("ablation_build"):
        pytest.skip("the LDS-DMA prototype kernel is only part of a -DDH_ABLATION build (DROID_HIP_ABLATION=1)")
    option("conv_halo", halo)
    option("conv_dma", dma)
    option("conv_halo2", halo2)
    """the raw convolution entry point against torch's fp32 conv2d: 1x1 / 3x3 / 7x7, multi-segment input, all tile configs"""
    from droid_amd.update import pack_conv, pack_conv_halo, EPI_LINEAR, EPI_RELU
    torch.manual_seed(0)
    cases = [(3, 12, 16, c) for c in [((128, 64, 8), 128, 3), ((200,), 64, 1), ((8,), 32, 7), ((128,), 576, 1), ((64, 64), 4, 3)]]
    # W == 64, H % 4 == 0: the 3x3 fast paths (image borders, 2 cout tiles, one to seven 64-channel chunks, segments
    # of unequal length, the 32- and 64-cout halo variants of the heads); (160, 96) has the channel count of the LDS-DMA
    # kernel but segments that are not multiples of 64 channels (generic loop)
    cases += [(2, 8, 64, c) for c in [((128, 64, 64), 128, 3), ((256,), 256, 3), ((64,), 256, 3), ((256,), 4, 3), ((128,), 1, 3), ((128,), 64, 3),
                                      ((128, 128, 128, 64), 256, 3), ((96, 32), 128, 3), ((160, 96), 128, 3)]]
    cases += [(3, 12, 64, ((64, 128), 128, 3))]
    cases += [(2, 8, 64, ((128,), 576, 1)), (1, 4, 64, ((128,), 320, 1))]      # conv1x1_c128_kernel: 9 / 5 cout tiles of 64 over a resident pixel tile
    for (N, H, W, (cins, cout, k)) in cases:
        xs = [torch.randn(N, H, W, c, device="cuda").half() for c in cins]
        wgt = torch.randn(cout, sum(cins), k, k, device="cuda") / (sum(cins) * k * k) ** 0.5
        bias = torch.randn(cout, device="cuda")
        wp, bp = pack_conv(wgt, bias)
        out = torch.empty(N, H, W, cout, device="cuda", dtype=torch.float16)
        db.conv2d_nhwc(xs, wp, pack_conv_halo(wgt), bp, k, k, cout, EPI_LINEAR, out, cout, None, None, None, None)
        x = torch.cat(xs, -1).float().permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(x, wgt.half().float(), bias, padding=k // 2).permute(0, 2, 3, 1)
        assert (out.float() - ref).abs().max() <= 1e-2 * max(1.0, ref.abs().max().item())


def test_conv_64_cout_layer_in_the_second_kernel_form(db, option):
    """conv3x3_halo64_kernel (option conv_halo64, the default since round 5): the flow encoder's 128 -> 64 layer as four waves of
    64 px x 64 couts with LDS-DMA weights, against torch's fp32 convolution and against the first halo kernel (conv_halo64 = 0;
    different accumulation order: one fp16 ulp), image borders and two segments included.  Then the launches that CANNOT take the
    kernel while the weights are packed for it (fp32 output; an image that is not 64 wide): they must fall to the generic loop
    (which reads the plain weight copy), never hand the padded layout to the first halo kernel."""
    from droid_amd.update import pack_conv, pack_conv_halo, EPI_LINEAR, EPI_RELU
    assert db.get_option("conv_halo64") == 1
    torch.manual_seed(1)
    for (N, H, W, cins, cout) in [(2, 8, 64, (128,), 64), (3, 12, 64, (64, 64), 64), (8, 48, 64, (128,), 64), (1, 4, 64, (32, 96), 48)]:
        xs = [torch.randn(N, H, W, c, device="cuda").half() for c in cins]
        wgt = torch.randn(cout, sum(cins), 3, 3, device="cuda") / (sum(cins) * 9) ** 0.5
        bias = torch.randn(cout, device="cuda")
        wp, bp = pack_conv(wgt, bias)
        x = torch.cat(xs, -1).float().permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(x, wgt.half().float(), bias, padding=1).permute(0, 2, 3, 1)
        for epi in (EPI_LINEAR, EPI_RELU):
            want = ref if epi == EPI_LINEAR else ref.clamp_min(0)
            outs = []
            for v in (0, 1):
                option("conv_halo64", v)
                wh = pack_conv_halo(wgt)
                assert wh.numel() == (128 if v else 64) * sum(cins) * 9          # padded halo2 layout / the first halo kernel's
                out = torch.empty(N, H, W, cout, device="cuda", dtype=torch.float16)
                db.conv2d_nhwc(xs, wp, wh, bp, 3, 3, cout, epi, out, cout, None, None, None, None)
                torch.cuda.synchronize()
                assert (out.float() - want).abs().max() <= 1e-2 * max(1.0, want.abs().max().item()), (v, N, H, cins, cout)
                outs.append(out.float())
            assert (outs[0] - outs[1]).abs().max() <= 2.0 ** -9 * max(1.0, want.abs().max().item())
    # ---- launches outside the kernel's domain with the padded layout in `weights_halo`
    option("conv_halo64", 1)
    for (N, H, W, f32_out) in [(2, 8, 64, True), (2, 8, 32, False), (1, 6, 64, False)]:
        xs = [torch.randn(N, H, W, 128, device="cuda").half()]
        wgt = torch.randn(64, 128, 3, 3, device="cuda") / (128 * 9) ** 0.5
        bias = torch.randn(64, device="cuda")
        wp, bp = pack_conv(wgt, bias)
        ref = torch.nn.functional.conv2d(xs[0].float().permute(0, 3, 1, 2), wgt.half().float(), bias, padding=1).permute(0, 2, 3, 1)
        out = torch.empty(N, H, W, 64, device="cuda", dtype=torch.float32 if f32_out else torch.float16)
        db.conv2d_nhwc(xs, wp, pack_conv_halo(wgt), bp, 3, 3, 64, EPI_LINEAR, out, 64, None, None, None, None, out_raw_f32=f32_out)
        torch.cuda.synchronize()
        assert (out.float() - ref).abs().max() <= 1e-2 * max(1.0, ref.abs().max().item()), (N, H, W, f32_out)


def test_context_term_in_the_accumulator_tile_layout_is_the_same_numbers(db, option):
    """round 5: the gates' per-frame context term is stored as the register tiles of conv3x3_halo2_kernel (option cinit_tiled, the
    default) -- [frame][pixel tile of 256][cout tile of 128][wave][a*2+b][q>>2][lane][q&3] -- and restored by the gate launches with
    16-byte loads.  (a) un-tiled on the host it EQUALS the pixel-major tensor element for element; (b) the update operator returns
    bit-identical results with either form (same fp32 start values, same MFMA order); (c) a slice of whole frames stays valid."""
    from droid_amd.update import UpdateModule
    from droid_amd.weights import deterministic_state_dict
    from oracle import update as oupd

    class _SD:
        def state_dict(self):
            return oupd.empty_state_dict()
    torch.manual_seed(5)
    upd = UpdateModule("cuda").load_state_dict(deterministic_state_dict(_SD(), seed=3))
    K, E, h, w = 3, 7, 12, 64
    inp_frames = torch.relu(torch.randn(K, h, w, 128, device="cuda")).half()
    assert db.get_option("cinit_tiled") == 1
    t = upd.context_term(inp_frames)
    p = upd.context_term(inp_frames, tiled=False)
    assert tuple(t.shape) == (K, h * w // 256, 3, 32768) and tuple(p.shape) == (K, h, w, 384)
    # (a) host-side un-tiling by the layout rule of include/droid_hip.h
    tt = t.view(K, h * w // 256, 3, 8, 2, 2, 4, 64, 4).cpu()            # [k, pt, ct, wave, a, b, j, lane, i]
    wave, a, b, j, lane, i = torch.meshgrid(*[torch.arange(n) for n in (8, 2, 2, 4, 64, 4)], indexing="ij")
    q = 4 * j + i
    pix = (wave & 3) * 64 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)          # pixel inside the 256-pixel tile
    co = (wave >> 2) * 64 + b * 32 + (lane & 31)                                       # cout inside the 128-cout tile
    pp = p.view(K, h * w // 256, 256, 3, 128).cpu()
    assert torch.equal(tt, pp[:, :, pix, :, co].permute(6, 7, 8, 0, 1, 2, 3, 4, 5))
    # (b) the operator with either form
    ii = torch.tensor([0, 0, 0, 1, 1, 2, 2], device="cuda")
    net0 = torch.tanh(torch.randn(E, h, w, 128, device="cuda")).half()
    corr = torch.randn(E, 196, h, w, device="cuda").half()
    flow = torch.zeros(E, h, w, 8, device="cuda", dtype=torch.float16); flow[..., :4] = torch.randn(E, h, w, 4, device="cuda").half()
    outs = []
    red = _glo_sums(upd, net0)                   # (reduced ONCE: the order of the reduction's atomics is free)
    for ctx in (t, p, None):
        n = net0.clone()
        r = upd.forward_nhwc(n, None, corr, flow, ii, inp_frames=inp_frames, inp_index=ii, ctx=ctx, glo_red=red)
        torch.cuda.synchronize()
        outs.append([x.clone() for x in r])
    option("cinit_tiled", 0)
    n = net0.clone()
    r = upd.forward_nhwc(n, None, corr, flow, ii, inp_frames=inp_frames, inp_index=ii, glo_red=red)
    torch.cuda.synchronize()
    outs.append([x.clone() for x in r])
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert torch.equal(x, y)
    # (c) frames 1..2 only, edges re-indexed
    option("cinit_tiled", 1)
    sel = ii >= 1
    n = net0[sel].clone()
    r = upd.forward_nhwc(n, None, corr[sel].contiguous(), flow[sel].contiguous(), ii[sel].contiguous(), inp_frames=inp_frames[1:], inp_index=(ii[sel] - 1).contiguous(), ctx=t[1:],
                         glo_red=red[sel].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(r[0], outs[0][0][sel]) and torch.equal(r[1], outs[0][1][sel]) and torch.equal(r[2], outs[0][2][sel])


def _glo_sums(upd, net):
    """the ConvGRU's global-context sums of `net` [E,h,w,128] by the stand-alone kernel -> [E,128] f32 (forward_nhwc(glo_red=...))"""
    from droid_amd.update import EPI_GLO
    red = torch.zeros(net.shape[0], 128, dtype=torch.float32, device=net.device)
    upd.params["gru_w"]([net], EPI_GLO, aux0=net, red=red)
    torch.cuda.synchronize()
    return red


def test_gates_on_64_cout_tiles_three_workgroups_per_cu_are_bit_identical(db, option):
    """round 6 (option conv_gate64, an A/B switch, off by default): the 3x3 / 128-cout layers -- the GRU gates with their epilogues and the
    accumulator-tile start values (bit 0), the relu layers (bit 1) -- through conv3x3_halo64_kernel on 64-cout tiles (three workgroups
    per CU) instead of conv3x3_halo2_kernel: the same MFMAs in the same k order on the same start values, so every output of the update
    operator is EQUAL bit for bit; also at the image borders and with a pixel-tile count that is not a multiple of 8."""
    from droid_amd.update import UpdateModule
    from droid_amd.weights import deterministic_state_dict
    from oracle import update as oupd

    class _SD:
        def state_dict(self):
            return oupd.empty_state_dict()
    torch.manual_seed(11)
    upd = UpdateModule("cuda").load_state_dict(deterministic_state_dict(_SD(), seed=4))
    assert db.get_option("conv_gate64") == 0
    for (K, E, h, w) in [(3, 7, 12, 64), (2, 5, 48, 64)]:
        inp_frames = torch.relu(torch.randn(K, h, w, 128, device="cuda")).half()
        ii = (torch.arange(E, device="cuda") * K // E).contiguous()
        net0 = torch.tanh(torch.randn(E, h, w, 128, device="cuda")).half()
        corr = torch.randn(E, 196, h, w, device="cuda").half()
        flow = torch.zeros(E, h, w, 8, device="cuda", dtype=torch.float16); flow[..., :4] = torch.randn(E, h, w, 4, device="cuda").half()
        outs = []
        red = _glo_sums(upd, net0)               # (reduced ONCE: the order of the reduction's atomics is free, two runs of it may differ in the last fp32 bit)
        for v in (0, 1, 2, 3):
            option("conv_gate64", v)
            for tiled in (True, False) if v in (0, 1) else (True,):
                n = net0.clone()
                ctx = upd.context_term(inp_frames, tiled=tiled)
                r = upd.forward_nhwc(n, None, corr, flow, ii, inp_frames=inp_frames, inp_index=ii, ctx=ctx, glo_red=red)
                torch.cuda.synchronize()
                outs.append([x.clone() for x in r])
        option("conv_gate64", 0)
        for other in outs[1:]:
            for x, y in zip(outs[0], other):
                assert torch.equal(x, y)


def test_two_pixel_tiles_per_workgroup_are_bit_identical(db, option):
    """round 6 (option conv_two_tiles): the relu layers with 128 input channels and the heads' first layer with TWO vertically adjacent
    pixel tiles per workgroup of conv3x3_halo2_kernel (the second tile's first fetches under the first tile's epilogue).  Same MFMAs per
    tile: every output EQUAL to the one-tile form -- an even and an ODD number of pixel tiles (the last workgroup then has one tile), tile
    pairs that straddle two images (3 tiles per image), image borders; and against torch's fp32 convolution on the first images."""
    from droid_amd.update import UpdateModule, pack_conv, pack_conv_halo, EPI_RELU, EPI_HEADS0
    from droid_amd.weights import deterministic_state_dict
    from oracle import update as oupd

    class _SD:
        def state_dict(self):
            return oupd.empty_state_dict()
    torch.manual_seed(31)
    upd = UpdateModule("cuda").load_state_dict(deterministic_state_dict(_SD(), seed=8))
    P = upd.params
    saved = db.get_option("conv_two_tiles")
    for (E, h) in [(683, 12), (172, 48)]:                      # 2049 (odd) / 2064 pixel tiles: at least 2048 take the two-tile form
        x = torch.tanh(torch.randn(E, h, 64, 128, device="cuda")).half()
        outs = []
        for v in (0, 1):
            option("conv_two_tiles", v)
            y = P["agg1"]([x], EPI_RELU)
            part = torch.empty(2, E * h // 4, 6, 64, 4, dtype=torch.float32, device="cuda")
            P["heads0"]([x], EPI_HEADS0, aux1=P["heads2_fused"][0], red=part)
            torch.cuda.synchronize()
            outs.append((y.clone(), part.clone()))
        option("conv_two_tiles", saved)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        sd = deterministic_state_dict(_SD(), seed=8)
        wgt, bias = sd["agg.conv1.weight"].cuda().float(), sd["agg.conv1.bias"].cuda().float()
        for sl in (slice(0, 2), slice(E - 2, E)):
            ref = torch.nn.functional.conv2d(x[sl].float().permute(0, 3, 1, 2), wgt.half().float(), bias, padding=1).permute(0, 2, 3, 1).clamp_min(0)
            assert (outs[1][0][sl].float() - ref).abs().max() <= 1e-2 * max(1.0, ref.abs().max().item())


def test_next_iterations_global_context_sums_from_inside_the_q_gate(db, option):
    """round 6 (dh_conv2d_nhwc_f16_ex3, option glo_fused): the ConvGRU starts with glo = mean(sigmoid(w(net)) * net) of the state the
    PREVIOUS iteration wrote (gru.py:23-24, :31).  forward_nhwc(glo_next=True) reduces the new state inside the q gate's launch:
    (a) those sums equal the stand-alone kernel's on the state that was written (same arithmetic per 256-pixel tile; the order of the
    per-tile atomics is free in both); (b) handing them to the next call gives the outputs of a call that recomputes them; (c) the state
    and every other output of the call that produced them are EQUAL to a call without the fused reduction; (d) both context forms
    (per-frame term / per-edge features); (e) a canvas image never fuses."""
    from droid_amd.update import UpdateModule, EPI_GLO
    from droid_amd.weights import deterministic_state_dict
    from oracle import update as oupd

    class _SD:
        def state_dict(self):
            return oupd.empty_state_dict()
    torch.manual_seed(21)
    upd = UpdateModule("cuda").load_state_dict(deterministic_state_dict(_SD(), seed=6))
    P = upd.params
    assert db.get_option("glo_fused") == 1
    for (K, E, h, w) in [(3, 7, 12, 64), (2, 9, 48, 64)]:
        assert upd.fuses_next_glo(h, w)
        inp_frames = torch.relu(torch.randn(K, h, w, 128, device="cuda")).half()
        ii = (torch.arange(E, device="cuda") * K // E).contiguous()
        net0 = torch.tanh(torch.randn(E, h, w, 128, device="cuda")).half()
        corr = [torch.randn(E, 196, h, w, device="cuda").half() for _ in range(2)]
        flow = torch.zeros(E, h, w, 8, device="cuda", dtype=torch.float16); flow[..., :4] = torch.randn(E, h, w, 4, device="cuda").half()
        for per_edge in (False, True):
            kw = dict(inp_frames=inp_frames, inp_index=ii) if not per_edge else {}
            inp = inp_frames[ii].contiguous() if per_edge else None
            red0 = _glo_sums(upd, net0)          # (the first iteration's sums, reduced ONCE: two runs of the reduction may differ in the last fp32 bit)
            # two iterations, plain
            n_a = net0.clone()
            r1a = [x.clone() for x in upd.forward_nhwc(n_a, inp, corr[0], flow, ii, glo_red=red0, **kw)]
            assert upd.last_glo is None
            r2a = [x.clone() for x in upd.forward_nhwc(n_a, inp, corr[1], flow, ii, **kw)]
            # two iterations, chained
            n_b = net0.clone()
            r1b = [x.clone() for x in upd.forward_nhwc(n_b, inp, corr[0], flow, ii, glo_red=red0, glo_next=True, **kw)]
            glo = upd.last_glo
            assert glo is not None and tuple(glo.shape) == (E, 128)
            for x, y in zip(r1a, r1b):                                                    # (c)
                assert torch.equal(x, y)
            red = torch.zeros(E, 128, device="cuda")
            P["gru_w"]([n_b], EPI_GLO, aux0=n_b, red=red)                                 # (a)
            torch.cuda.synchronize()
            assert (glo - red).abs().max().item() <= 1e-5 * max(1.0, red.abs().max().item())
            r2b = [x.clone() for x in upd.forward_nhwc(n_b, inp, corr[1], flow, ii, glo_red=glo, glo_next=True, **kw)]
            assert upd.last_glo is not None and upd.last_glo is not glo
            for x, y in zip(r2a, r2b):                                                    # (b)
                assert (x.float() - y.float()).abs().max().item() <= 2.0 ** -9 * max(1.0, x.float().abs().max().item())
            option("glo_fused", 0)
            upd.forward_nhwc(net0.clone(), inp, corr[0], flow, ii, glo_next=True, **kw)
            assert upd.last_glo is None
            option("glo_fused", 1)
    # (e) 30 x 40 runs on a canvas: the reduction stays a kernel of its own (the padding is re-zeroed after the gate)
    E, h, w = 3, 30, 40
    net = torch.tanh(torch.randn(E, h, w, 128, device="cuda")).half()
    inp = torch.relu(torch.randn(E, h, w, 128, device="cuda")).half()
    flow = torch.zeros(E, h, w, 8, device="cuda", dtype=torch.float16)
    upd.forward_nhwc(net, inp, torch.randn(E, 196, h, w, device="cuda").half(), flow, torch.arange(E, device="cuda"), glo_next=True)
    assert upd.last_glo is None


def test_operator_without_the_upmask_head(db):
    """forward_nhwc(want_upmask=False) -- what FactorGraph asks for when it does not upsample (the head's only reader is
    DepthVideo.upsample; the reference computes and drops it): upmask is None, every other output EQUAL to the full call."""
    from droid_amd.update import UpdateModule
    from droid_amd.weights import deterministic_state_dict
    from oracle import update as oupd

    class _SD:
        def state_dict(self):
            return oupd.empty_state_dict()
    torch.manual_seed(41)
    upd = UpdateModule("cuda").load_state_dict(deterministic_state_dict(_SD(), seed=2))
    for (K, E, h, w) in [(2, 5, 12, 64), (2, 3, 30, 40)]:
        inp_frames = torch.relu(torch.randn(K, h, w, 128, device="cuda")).half()
        ii = (torch.arange(E, device="cuda") * K // E).contiguous()
        net0 = torch.tanh(torch.randn(E, h, w, 128, device="cuda")).half()
        corr = torch.randn(E, 196, h, w, device="cuda").half()
        flow = torch.zeros(E, h, w, 8, device="cuda", dtype=torch.float16); flow[..., :4] = torch.randn(E, h, w, 4, device="cuda").half()
        red = _glo_sums(upd, net0) if w == 64 else None
        full = upd.forward_nhwc(net0.clone(), None, corr, flow, ii, inp_frames=inp_frames, inp_index=ii, glo_red=red)
        full = [x.clone() for x in full]
        lean = upd.forward_nhwc(net0.clone(), None, corr, flow, ii, inp_frames=inp_frames, inp_index=ii, glo_red=red, want_upmask=False)
        torch.cuda.synchronize()
        if w == 64:
            assert lean[4] is None and full[4] is not None
            for x, y in zip(full[:4], lean[:4]):
                assert torch.equal(x, y)
        else:                                  # canvas sizes keep the head (their path is not the one FactorGraph's large graphs take)
            assert lean[4] is not None


def test_conv7x7_on_four_channels_and_global_context_kernels(db):
    """the two single-purpose kernels of the update operator against torch: flow_encoder.0 (7x7 on the 4 motion
    channels, droid_net.py:89) and the ConvGRU's global-context reduction mean(sigmoid(w(net)) * net) (gru.py:23-24)"""
    from droid_amd.update import pack_conv, pack_conv_7x7_c4, EPI_RELU, EPI_GLO
    torch.manual_seed(3)
    N, H, W = 3, 8, 64
    x = torch.zeros(N, H, W, 8, device="cuda", dtype=torch.float16)
    x[..., :4] = (4 * torch.randn(N, H, W, 4, device="cuda")).half()
    wgt = torch.randn(128, 4, 7, 7, device="cuda") / 14.0
    bias = torch.randn(128, device="cuda")
    wp, bp = pack_conv(wgt, bias, 8)
    out = torch.empty(N, H, W, 128, device="cuda", dtype=torch.float16)
    db.conv2d_nhwc([x], wp, pack_conv_7x7_c4(wgt), bp, 7, 7, 128, EPI_RELU, out, 128, None, None, None, None)
    ref = torch.relu(torch.nn.functional.conv2d(x[..., :4].float().permute(0, 3, 1, 2), wgt.half().float(), bias, padding=3)).permute(0, 2, 3, 1)
    assert (out.float() - ref).abs().max() <= 2.0 ** -9 * max(1.0, ref.abs().max().item())
    gen = torch.empty_like(out)                          # generic loop on the same inputs
    db.conv2d_nhwc([x], wp, None, bp, 7, 7, 128, EPI_RELU, gen, 128, None, None, None, None)
    assert (out.float() - gen.float()).abs().max() <= 2.0 ** -9 * max(1.0, ref.abs().max().item())
    # the two forms of the stem kernel (option conv_c7_split: 64-cout halves, four workgroups per CU): the same MFMAs -> EQUAL
    saved = db.get_option("conv_c7_split")
    both = []
    for v in (0, 1):
        db.set_option("conv_c7_split", v)
        o = torch.empty_like(out)
        db.conv2d_nhwc([x], wp, pack_conv_7x7_c4(wgt), bp, 7, 7, 128, EPI_RELU, o, 128, None, None, None, None)
        torch.cuda.synchronize()
        both.append(o)
    db.set_option("conv_c7_split", saved)
    assert torch.equal(both[0], both[1]) and torch.equal(both[0], out if saved == 0 else both[saved])
    # round 6: outputs of 64 MB and more leave the stem and the 1x1 128 -> 576 upmask head with NON-TEMPORAL stores (option conv_nt_out;
    # inline-asm stores the compiler does not track): the same bytes as the plain stores, also right before the kernel ends
    saved_nt = db.get_option("conv_nt_out")
    xb = torch.zeros(96, 48, W, 8, device="cuda", dtype=torch.float16)              # 96 x 48 x 64 x 128 x 2 B = 75.5 MB of output
    xb[..., :4] = (4 * torch.randn(96, 48, W, 4, device="cuda")).half()
    a128 = torch.relu(torch.randn(24, 48, W, 128, device="cuda")).half()            # 24 x 48 x 64 x 576 x 2 B = 85 MB
    w576 = torch.randn(576, 128, 1, 1, device="cuda") / 11.0
    wp576, bp576 = pack_conv(w576, torch.randn(576, device="cuda"))
    outs = []
    for v in (0, 1):
        db.set_option("conv_nt_out", v)
        o7 = torch.full((96, 48, W, 128), -7.0, device="cuda", dtype=torch.float16)
        db.conv2d_nhwc([xb], wp, pack_conv_7x7_c4(wgt), bp, 7, 7, 128, EPI_RELU, o7, 128, None, None, None, None)
        o1 = torch.full((24, 48, W, 576), -7.0, device="cuda", dtype=torch.float16)
        db.conv2d_nhwc([a128], wp576, None, bp576, 1, 1, 576, 0, o1, 576, None, None, None, None)
        torch.cuda.synchronize()
        outs.append((o7, o1))
    db.set_option("conv_nt_out", saved_nt)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[1][0].min()) >= 0.0 and not bool((outs[1][1] == -7.0).all())
    # the persistent form (option conv_c7_pp: one 16-wave workgroup per CU, two wave groups alternating between multiplying a tile and
    # parking / storing the previous one): EQUAL -- 6 tiles (two per workgroup), 2 101 tiles (odd; eight or nine per workgroup), 1 tile
    # ... and the form with sixteen 32 px x 64 cout waves per workgroup (option conv_c7_w16): EQUAL as well
    saved_pp, saved_w16 = db.get_option("conv_c7_pp"), db.get_option("conv_c7_w16")
    for (n2, h2) in [(N, H), (191, 44), (1, 4)]:
        x2 = torch.zeros(n2, h2, W, 8, device="cuda", dtype=torch.float16)
        x2[..., :4] = (4 * torch.randn(n2, h2, W, 4, device="cuda")).half()
        res = []
        for (pp, w16) in ((0, 0), (1, 0), (0, 1)):
            db.set_option("conv_c7_pp", pp); db.set_option("conv_c7_w16", w16)
            o = torch.full((n2, h2, W, 128), -7.0, device="cuda", dtype=torch.float16)
            db.conv2d_nhwc([x2], wp, pack_conv_7x7_c4(wgt), bp, 7, 7, 128, EPI_RELU, o, 128, None, None, None, None)
            torch.cuda.synchronize()
            res.append(o)
        db.set_option("conv_c7_pp", saved_pp); db.set_option("conv_c7_w16", saved_w16)
        assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2]), (n2, h2)
    # global context
    net = torch.tanh(torch.randn(N, H, W, 128, device="cuda")).half()
    w1 = torch.randn(128, 128, 1, 1, device="cuda") / 11.0
    b1 = torch.randn(128, device="cuda")
    wp, bp = pack_conv(w1, b1)
    red = torch.zeros(N, 128, device="cuda")
    db.conv2d_nhwc([net], wp, None, bp, 1, 1, 128, EPI_GLO, None, 0, None, net, None, red)
    g = torch.sigmoid(torch.nn.functional.conv2d(net.float().permute(0, 3, 1, 2), w1.half().float(), b1)).half().float()
    ref = (g * net.float().permute(0, 3, 1, 2)).half().float().sum((2, 3))
    assert (red - ref).abs().max() <= 2e-3 * max(1.0, ref.abs().max().item())
    db.set_option("conv_halo", 0)                        # the generic loop's epilogue on the same inputs
    try:
        red2 = torch.zeros(N, 128, device="cuda")
        db.conv2d_nhwc([net], wp, None, bp, 1, 1, 128, EPI_GLO, None, 0, None, net, None, red2)
    finally:
        db.set_option("conv_halo", 1)
    assert (red - red2).abs().max() <= 2e-3 * max(1.0, ref.abs().max().item())


def test_native_pyramid_channel_last_lookup_equals_reference_layout(db):
    """corr_pyramid_lookup_nhwc == corr_pyramid_lookup up to the documented channel permutation"""
    from droid_amd.corr import CorrBlock
    from droid_amd.update import corr_channel_map
    torch.manual_seed(5)
    for (E, h, w) in [(3, 16, 16), (2, 48, 64)]:
        f1 = torch.randn(1, E, 128, h, w, device="cuda").half()
        f2 = torch.randn(1, E, 128, h, w, device="cuda").half()
        coords = dev(_smooth_coords(np.random.default_rng(2), E, h, w))[None]
        blk = CorrBlock(f1, f2)
        a = blk(coords)[0]                        # [E,196,h,w]
        b = blk.lookup_nhwc(coords)               # [4,E,h,w,56]
        b = b.permute(1, 2, 3, 0, 4).reshape(E, h, w, 224)
        m = corr_channel_map().cuda()
        # same taps, same interpolation; the two template instantiations may contract a*b+c differently (-ffp-contract):
        # equal up to one fp16 rounding of the result
        x, y = b[..., m >= 0].float(), a.permute(0, 2, 3, 1)[..., m[m >= 0]].float()
        assert (x - y).abs().max() <= 2.0 ** -10 * y.abs().max() and (x != y).float().mean() < 1e-4
        assert torch.count_nonzero(b[..., m < 0]) == 0


def test_update_operator_edge_segments_follow_the_edge_list(db):
    """UpdateModule.segments caches GraphAgg's grouping per edge list: a new tensor, a tensor modified in place, and a
    tensor reallocated at the same address must all be regrouped"""
    from droid_amd.update import UpdateModule
    upd = UpdateModule()
    def groups(ii):
        order, off = upd.segments(ii)
        o, f = order.cpu().tolist(), off.cpu().tolist()
        return sorted(sorted(o[f[k]:f[k + 1]]) for k in range(len(f) - 1))
    expect = lambda ii: sorted(sorted(i for i, v in enumerate(ii) if v == u) for u in set(ii))
    a = [0, 0, 1, 2, 2, 2]
    ii = torch.tensor(a, device="cuda")
    assert groups(ii) == expect(a)
    assert upd.segments(ii)[0] is upd.segments(ii)[0]                      # cached
    ii[1] = 1                                                              # in place
    assert groups(ii) == expect([0, 1, 1, 2, 2, 2])
    del ii
    b = [3, 1, 1, 1, 0, 3]
    jj = torch.tensor(b, device="cuda")                                    # may land on the freed address
    assert groups(jj) == expect(b)


def test_corr0_on_reference_layout_features(db):
    """droid_backends.corr0_nchw: corr_encoder.0 (1x1, 196 -> 128, relu, droid_net.py:83-86) reading [E,196,h,w] directly
    (in-register tile transpose) == fp32 convolution of the same fp16 operands, and == the channel-last implicit-GEMM path"""
    torch.manual_seed(11)
    for (E, h, w) in [(3, 16, 64), (5, 8, 16), (2, 48, 64)]:
        x = (torch.randn(E, 196, h, w, device="cuda") * 2).half()
        wgt = (torch.randn(128, 196, device="cuda") * 0.1).half()
        bias = torch.randn(128, device="cuda")
        wp = torch.zeros(128, 208, device="cuda", dtype=torch.half); wp[:, :196] = wgt
        out = db.corr0_nchw(x, wp, bias)                                   # [E,h,w,128] f16
        ref = torch.relu(torch.einsum("ekhw,ck->ehwc", x.float(), wgt.float()) + bias)
        assert out.shape == (E, h, w, 128) and out.dtype == torch.half
        assert (out.float() - ref).abs().max() <= 2.0 ** -9 * ref.abs().max()
    with pytest.raises(RuntimeError):
        db.corr0_nchw(torch.zeros(1, 196, 12, 16, device="cuda").half(), wp, bias)      # 192 pixels: not a multiple of 128


def test_lookup_fused_with_first_encoder_layer(db, option):
    """droid_backends.corr_pyramid_lookup_corr0 (lookup + Conv2d(196,128,1) + ReLU of droid_net.py:96-100 in one kernel) ==
    the 1x1 layer in fp32 on the fp16 samples that corr_pyramid_lookup stores, for windows inside, across and outside the
    image, at all three widths, and with more strips than persistent workgroups (E*h/8 > number of CUs)"""
    from droid_amd.update import pack_corr0_fused
    torch.manual_seed(5)
    rng = np.random.default_rng(8)
    for (E, h, w) in [(3, 16, 64), (4, 8, 16), (3, 16, 32), (1, 48, 64), (120, 48, 64)]:
        f1 = torch.randn(E, 128, h, w, device="cuda").half()
        f2 = torch.randn(E, 128, h, w, device="cuda").half()
        pyr = db.corr_pyramid_build(f1, f2)
        coords = dev(_smooth_coords(rng, E, h, w))
        coords[E // 2] += 7.3                                             # one edge with windows leaving the image
        wgt = (torch.randn(128, 196, device="cuda") * 0.05)
        bias = torch.randn(128, device="cuda") * 0.3
        wpk = pack_corr0_fused(wgt)
        samples = db.corr_pyramid_lookup(pyr, coords)                     # [E,196,h,w] f16
        out = db.corr_pyramid_lookup_corr0(pyr, coords, wpk, bias)
        torch.cuda.synchronize()
        assert out.shape == (E, h, w, 128) and out.dtype == torch.half
        ref = torch.empty(E, h, w, 128, device="cuda")
        for e0 in range(0, E, 8):
            ref[e0:e0 + 8] = torch.relu(torch.einsum("ekhw,ck->ehwc", samples[e0:e0 + 8].float(), wgt.half().float()) + bias)
        assert ref.abs().max() > 1.0
        err = (out.float() - ref).abs().max().item()
        assert err <= 2.0 ** -9 * ref.abs().max().item(), (E, h, w, err)
        assert torch.equal(out, db.corr_pyramid_lookup_corr0(pyr, coords, wpk, bias))       # deterministic
        if w == 64:
            # the synchronous twin of the kernel (lookup_mode 6: every tap batch is waited for where it is issued) computes the same
            # sums in the same order: bit-identical output <=> the product kernel's explicit vmcnt bookkeeping let no register be
            # read, copied or spilled while its load was in flight on this launch
            option("lookup_mode", 6)
            sync = db.corr_pyramid_lookup_corr0(pyr, coords, wpk, bias)
            option("lookup_mode", 0)
            assert torch.equal(out, sync)
            del sync
            # the interpolation through v_fma_mix_f32 / v_fma_mixlo_f16 (default) performs the same fp32 operations in the same
            # order as the form with the fp16 <-> fp32 conversions spelled out: bit-identical
            option("lookup_mix", 0)
            plain = db.corr_pyramid_lookup_corr0(pyr, coords, wpk, bias)
            option("lookup_mix", 1)
            assert torch.equal(out, plain)
            del plain
            # the other refill schedule of the tap registers (window row by window row instead of by half level): the same loads and
            # the same arithmetic under different vmcnt bookkeeping
            if db.get_option("ablation_build"):              # (DROID_HIP_TEST_ABLATION=1; the release build has one schedule)
                option("lookup_fill", 1)
                other = db.corr_pyramid_lookup_corr0(pyr, coords, wpk, bias)
                option("lookup_fill", 0)
                assert torch.equal(out, other)
                del other
        del pyr, samples, out, ref


def test_update_operator_same_result_from_both_correlation_layouts(db):
    """forward_nhwc on the reference-layout features [E,196,h,w] (corr0_nchw) and on the channel-last level-planar
    features (implicit GEMM): same operator, results within one fp16 rounding of the first layer"""
    from droid_amd.update import UpdateModule
    torch.manual_seed(3)
    E, h, w = 6, 16, 64
    from oracle import update as oupd
    from droid_amd.weights import deterministic_state_dict
    upd = UpdateModule().load_state_dict(deterministic_state_dict(_SD(oupd.empty_state_dict()), seed=7))
    corr = (torch.randn(E, 196, h, w, device="cuda")).half()
    net0 = torch.tanh(torch.randn(E, h, w, 128, device="cuda")).half()
    inp = torch.relu(torch.randn(E, h, w, 128, device="cuda")).half()
    flow = torch.zeros(E, h, w, 8, device="cuda").half(); flow[..., :4] = torch.randn(E, h, w, 4, device="cuda").half()
    ii = torch.tensor([0, 0, 1, 1, 2, 2], device="cuda")
    assert upd.wants_reference_layout_corr(h, w)
    outs = []
    for feats in (corr, upd.corr_to_nhwc(corr)):
        net = net0.clone()
        n, d, wt, eta, um = upd.forward_nhwc(net, inp, feats, flow, ii)
        outs.append([t.float().clone() for t in (n, d, wt, eta, um)])
    for a, b in zip(*outs):
        assert (a - b).abs().max() <= 2.0 ** -8 * max(1.0, b.abs().max().item())


# ------------------------------------------------------------------------------------------ lietorch / torch_scatter drop-ins
def test_lietorch_dropin_matches_oracle_and_reference_call_pattern(db):
    """the SE3 surface the reference's projective_ops.py:165-198 uses, through the `lietorch` package of this repo"""
    import lietorch
    from lietorch import SE3
    rng = np.random.default_rng(3)
    xi = rng.normal(0, 0.3, (1, 9, 6)).astype(np.float32)
    G = SE3.exp(dev(xi))                                               # [1,9]
    t, q = ose3.se3_exp(xi.astype(np.float64))
    assert np.abs(G.data.cpu().numpy() - ose3.pose_join(t, q)).max() < 1e-5
    ii = torch.tensor([0, 1, 2, 5, 8], device="cuda"); jj = torch.tensor([1, 0, 4, 5, 2], device="cuda")
    Gij = G[:, jj] * G[:, ii].inv()                                    # projective_ops.py:174
    ti, qi = t[0][ii.cpu().numpy()], q[0][ii.cpu().numpy()]
    tj, qj = t[0][jj.cpu().numpy()], q[0][jj.cpu().numpy()]
    tinv, qinv = ose3.se3_inv(ti, qi)
    tr, qr = ose3.se3_mul(tj, qj, tinv, qinv)
    assert np.abs(Gij.data[0].cpu().numpy() - ose3.pose_join(tr, qr)).max() < 1e-5
    X = dev(rng.normal(0, 1, (1, 5, 6, 7, 4)).astype(np.float32))
    Y = Gij[:, :, None, None] * X                                      # projective_ops.py:87 (broadcast over pixels)
    ref = ose3.se3_act(tr[:, None, None], qr[:, None, None], X[0].cpu().numpy().astype(np.float64))
    assert np.abs(Y[0].cpu().numpy() - ref).max() < 1e-5
    J = dev(rng.normal(0, 1, (1, 5, 6, 7, 2, 6)).astype(np.float32))
    A = Gij[:, :, None, None, None].adjT(J)                            # projective_ops.py:191
    refA = ose3.se3_adjT(tr[:, None, None, None], qr[:, None, None, None], J[0].cpu().numpy().astype(np.float64))
    assert np.abs(A[0].cpu().numpy() - refA).max() < 1e-4
    dxi = dev(rng.normal(0, 0.05, (1, 9, 6)).astype(np.float32))
    R = G.retr(dxi)                                                    # geom/ba.py:26-28
    t2, q2 = ose3.se3_retr(dxi.cpu().numpy().astype(np.float64), t, q)
    assert np.abs(R.data.cpu().numpy() - ose3.pose_join(t2, q2)).max() < 1e-5
    assert lietorch.cat([G, R], 1).shape == (1, 18) and SE3.IdentityLike(G).data[..., 6].min() == 1


def test_torch_scatter_dropin(db):
    import torch_scatter
    x = torch.randn(7, 5, device="cuda"); ix = torch.tensor([0, 2, 2, 1, 0, 2, 3], device="cuda")
    s = torch_scatter.scatter_sum(x, ix, dim=0); m = torch_scatter.scatter_mean(x, ix, dim=0)
    for k in range(4):
        assert torch.allclose(s[k], x[ix == k].sum(0), atol=1e-6)
        assert torch.allclose(m[k], x[ix == k].mean(0), atol=1e-6)


# ------------------------------------------------------------------------------------------ MFMA alt-correlation
@pytest.mark.parametrize("kind", ["smooth", "random"])
def test_altcorr_mfma_block_vs_oracle_and_reference_layout_path(db, kind):
    """AltCorrBlock (channel-last MFMA kernel; smooth flow = tile path, random flow = per-lane path, image borders in
    both) vs the oracle's alt lookup and vs the reference-layout kernel on identical inputs, all 4 levels."""
    from droid_amd.corr import AltCorrBlock
    rng = np.random.default_rng(11)
    N, C, H, W = 4, 128, 16, 24
    fm = rng.standard_normal((1, N, C, H, W)).astype(np.float16)
    ii = np.array([0, 1, 3, 2, 2]); jj = np.array([1, 0, 3, 0, 3])
    M = len(ii)
    if kind == "smooth":
        c = _smooth_coords(rng, M, H, W, amp=5.0)
    else:
        c = np.stack([rng.uniform(-4, W + 3, (M, H, W)), rng.uniform(-4, H + 3, (M, H, W))], -1).astype(np.float32)
    blk = AltCorrBlock(dev(fm))
    assert blk.mfma
    out = blk(dev(c)[None], dev(ii), dev(jj))[0].float().cpu().numpy()               # [M,196,H,W]
    ref = ocorr.alt_block_lookup(fm.astype(np.float32), c[None], ii, jj, 3, 4, pool_dtype=np.float16)[0]
    assert out.shape == ref.shape == (M, 196, H, W)
    assert np.abs(out - ref).max() <= 2.0 ** -9 * np.abs(ref).max()
    blk.mfma = False                                                                  # reference-layout kernel
    out2 = blk(dev(c)[None], dev(ii), dev(jj))[0].float().cpu().numpy()
    assert np.abs(out - out2).max() <= 2.0 ** -9 * np.abs(ref).max()


# ------------------------------------------------------------------------------------------ full-size BA properties
def test_ba_config_c3_full_size_properties(db):
    """BASELINE configs[2] (512 keyframes / 4096 edges / 48x64, global-BA damping): too large for the Python oracle, so
    the size-independent properties are checked: the weighted reprojection cost falls under repeated BA, quaternions
    stay unit, fixed frames stay fixed, everything is finite, and the 3066-unknown solve did not fall back to dx = 0."""
    g = syn.make_graph("C3")
    N = g["n_frames"]
    poses = dev(g["poses"]); disps = dev(g["disps"])
    args = [dev(g[k]) for k in ("intrinsics", "disps_sens", "targets", "weights", "eta", "ii", "jj")]
    cost = lambda: oba.reprojection_cost(poses.cpu().numpy(), disps.cpu().numpy(), g["intrinsics"], g["targets"],
                                         g["weights"], g["ii"], g["jj"])
    costs = [cost()]
    for _ in range(2):
        dx, dz = db.ba(poses, disps, *args, 1, N, 2, g["lm"], g["ep"], False)
        disps.clamp_(min=0.001)
        costs.append(cost())
        assert torch.isfinite(dx).all() and torch.isfinite(dz).all() and dx.abs().max() > 0
    assert costs[1] < 0.5 * costs[0] and costs[2] <= 1.01 * costs[1]
    p = poses.cpu().numpy()
    assert np.isfinite(p).all() and np.isfinite(disps.cpu().numpy()).all()
    assert np.abs(np.linalg.norm(p[:, 3:], axis=-1) - 1).max() < 1e-4
    assert np.array_equal(p[0], g["poses"][0])


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(3, 128, 30, 40), (2, 128, 48, 64), (2, 40, 9, 13)])
def test_reference_layout_volume_build_and_pooling(db, dtype, shape):
    """corr_volume_build / corr_volume_pool (any image size: TUM's 30x40, odd sizes, C not a multiple of 16) against
    CorrBlock.corr + F.avg_pool2d of the reference formulation (modules/corr.py:32-38,63-71) evaluated by torch"""
    E, C, h, w = shape
    g = torch.Generator(device="cuda").manual_seed(E * h + w)
    f1 = torch.randn(E, C, h, w, device="cuda", generator=g).to(dtype)
    f2 = torch.randn(E, C, h, w, device="cuda", generator=g).to(dtype)
    vol = db.corr_volume_build(f1, f2)
    want = torch.matmul((f1.float().reshape(E, C, h * w) / 4.0).transpose(1, 2), f2.float().reshape(E, C, h * w) / 4.0)
    assert vol.shape == (E, h, w, h, w) and vol.dtype == dtype
    tol = 2.0 ** -10 if dtype == torch.float16 else 1e-5
    assert (vol.float().reshape(E, h * w, h * w) - want).abs().max().item() <= tol * want.abs().max().item()
    cur = vol
    for _ in range(3):
        nxt = db.corr_volume_pool(cur)
        ref = torch.nn.functional.avg_pool2d(cur.float().reshape(-1, 1, cur.shape[-2], cur.shape[-1]), 2, stride=2)
        assert nxt.shape[-2:] == ref.shape[-2:] and nxt.shape[:3] == (E, h, w)
        assert (nxt.float().reshape(ref.shape) - ref).abs().max().item() <= (2.0 ** -10 if dtype == torch.float16 else 1e-6) * max(1.0, ref.abs().max().item())
        cur = nxt
    # the block built on them serves lookups at this size (the product path for images outside the MI355X pyramid layout)
    from droid_amd.corr import CorrBlockRef
    from oracle import corr as ocorr
    if C == 128:
        blk = CorrBlockRef(f1[None], f2[None])
        rng = np.random.default_rng(5)
        cc = np.stack([rng.uniform(-2, w + 1, (E, h, w)), rng.uniform(-2, h + 1, (E, h, w))], -1).astype(np.float32)
        got = blk(torch.as_tensor(cc).cuda()[None])[0].float().cpu().numpy()
        o = ocorr.corr_block_lookup(ocorr.corr_pyramid(f1.float().cpu().numpy(), f2.float().cpu().numpy(), 4), cc, 3)
        assert np.abs(got - o).max() <= (2.0 ** -8 if dtype == torch.float16 else 1e-4) * np.abs(o).max()


def test_update_forward_leaves_a_channel_last_hidden_state_untouched(db):
    """UpdateModule.forward (the reference's interface, droid_net.py:111) must not modify its arguments: a hidden state that
    already lies channel-last in memory (what this library's encoders hand to MotionFilter) makes the layout conversion a
    view, and forward_nhwc updates its hidden state in place"""
    from oracle import update as oupd
    from droid_amd.update import UpdateModule
    from droid_amd.weights import deterministic_state_dict
    mod = UpdateModule().load_state_dict(deterministic_state_dict(_SD(oupd.empty_state_dict()), seed=7))
    g = torch.Generator(device="cuda").manual_seed(1)
    net = torch.tanh(torch.randn(1, 2, 16, 64, 128, device="cuda", generator=g)).half().permute(0, 1, 4, 2, 3)   # [1,E,128,h,w], channel-last memory
    inp = torch.relu(torch.randn(1, 2, 16, 64, 128, device="cuda", generator=g)).half().permute(0, 1, 4, 2, 3)
    corr = torch.randn(1, 2, 196, 16, 64, device="cuda", generator=g).half()
    before = net.clone()
    out1 = mod(net, inp, corr)[0].clone()
    assert torch.equal(net, before)
    out2 = mod(net, inp, corr)[0]
    # (not torch.equal: the global-context reduction adds its per-tile sums with atomics, whose order -- and so the last fp32 bit -- is free)
    assert (out1.float() - out2.float()).abs().max().item() <= 2.0 ** -9


@pytest.mark.parametrize("variant", ["conv_halo3", "conv_halo4"])
def test_conv_512_pixel_tile_form_is_bit_identical(db, option, variant):
    """conv3x3_halo3_kernel (option conv_halo3: 8-row / 512-pixel tile, 128 x 64 per wave, one workgroup per CU) and
    conv3x3_halo4_kernel (conv_halo4: the production tile with four 64 x 128 waves) run the same k order per output element as
    conv3x3_halo2_kernel: raw convolutions (relu, multi-segment input, two cout tiles, several images, image borders) and the
    WHOLE update operator (gates with accumulator start values and GRU epilogues, fused heads) must be equal bit for bit with
    the option on and off."""
    if not db.get_option("ablation_build"):
        pytest.skip("measurement kernels of the -DDH_ABLATION build (DROID_HIP_TEST_ABLATION=1 runs them; profiles/r04_d_conv_halo3_ab.json)")
    from oracle import update as oupd
    from droid_amd.update import pack_conv, pack_conv_halo, EPI_RELU, UpdateModule
    from droid_amd.weights import deterministic_state_dict
    torch.manual_seed(5)
    for (n, h, cins, cout) in [(3, 8, (128, 128, 64), 256), (2, 16, (128,), 128), (1, 48, (32, 96), 128)]:
        xs = [torch.randn(n, h, 64, c, device="cuda").half() for c in cins]
        wgt = torch.randn(cout, sum(cins), 3, 3, device="cuda") / (sum(cins) * 9) ** 0.5
        wp, bp = pack_conv(wgt, torch.randn(cout, device="cuda"))
        wh = pack_conv_halo(wgt)
        outs = []
        for v in (0, 1):
            option(variant, v)
            out = torch.zeros(n, h, 64, cout, device="cuda", dtype=torch.float16)
            db.conv2d_nhwc(xs, wp, wh, bp, 3, 3, cout, EPI_RELU, out, cout, None, None, None, None)
            torch.cuda.synchronize()
            outs.append(out)
        ref = torch.nn.functional.conv2d(torch.cat(xs, -1).float().permute(0, 3, 1, 2), wgt.half().float(), bp[:cout], padding=1).permute(0, 2, 3, 1).clamp_min(0)
        assert (outs[1].float() - ref).abs().max().item() <= 2.0 ** -8 * max(1.0, ref.abs().max().item())
        assert torch.equal(outs[0], outs[1])

    class _SD:
        def state_dict(self):
            return oupd.empty_state_dict()
    mod = UpdateModule("cuda").load_state_dict(deterministic_state_dict(_SD(), seed=3))
    E, h, w = 6, 16, 64
    net = torch.tanh(torch.randn(E, h, w, 128, device="cuda")).half()
    inp_frames = torch.relu(torch.randn(3, h, w, 128, device="cuda")).half()
    c0 = torch.relu(torch.randn(E, h, w, 128, device="cuda")).half()
    flow = torch.zeros(E, h, w, 8, device="cuda", dtype=torch.float16); flow[..., :4] = (4.0 * torch.randn(E, h, w, 4, device="cuda")).clamp(-64, 64).half()
    ii = torch.tensor([0, 0, 1, 1, 2, 2], device="cuda")
    res = []
    for v in (0, 1):
        option(variant, v)
        r = mod.forward_nhwc(net.clone(), None, None, flow, ii, inp_frames=inp_frames, inp_index=ii, corr0=c0)
        torch.cuda.synchronize()
        res.append([t.clone() for t in r])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_conv_winograd_prototype(db):
    """conv3x3_wino_kernel (Winograd F(2,3) along x, opt-in prototype) against torch's fp32 convolution and against the direct
    kernel on the same operands: multi-segment input, two cout tiles, image borders, relu epilogue; and the two gate
    convolutions of the update operator (accumulator start values through the transform's null space, GRU epilogues)
    against the autocast oracle at the operator's 2^-9."""
    if not db.get_option("ablation_build"):
        pytest.skip("the Winograd prototype kernel is only part of a -DDH_ABLATION build (DROID_HIP_ABLATION=1)")
    from oracle import update as oupd
    from droid_amd.update import pack_conv, pack_conv_halo, pack_conv_wino, EPI_RELU, LAYOUT_WINO, UpdateModule
    from droid_amd.weights import deterministic_state_dict
    torch.manual_seed(3)
    for (N, H, cins, cout) in ((3, 8, (128, 128, 64), 256), (2, 12, (64,), 128)):
        xs = [torch.randn(N, H, 64, c, device="cuda").half() for c in cins]
        wgt = torch.randn(cout, sum(cins), 3, 3, device="cuda") / (sum(cins) * 9) ** 0.5
        bias = torch.randn(cout, device="cuda")
        wp, bp = pack_conv(wgt, bias)
        ref = torch.nn.functional.conv2d(torch.cat(xs, -1).float().permute(0, 3, 1, 2), wgt.half().float(), bias, padding=1).permute(0, 2, 3, 1).clamp_min(0)
        out_d = torch.empty(N, H, 64, cout, device="cuda", dtype=torch.float16); out_w = torch.empty_like(out_d)
        db.conv2d_nhwc(xs, wp, pack_conv_halo(wgt), bp, 3, 3, cout, EPI_RELU, out_d, cout, None, None, None, None)
        db.conv2d_nhwc(xs, wp, pack_conv_wino(wgt), bp, 3, 3, cout, EPI_RELU, out_w, cout, None, None, None, None, weights_layout=LAYOUT_WINO)
        torch.cuda.synchronize()
        scale = ref.abs().max().item()
        e_d = (out_d.float() - ref).abs().max().item(); e_w = (out_w.float() - ref).abs().max().item()
        assert e_d <= 2.0 ** -10 * scale + 2.0 ** -11                 # direct: one fp16 rounding of the result
        assert e_w <= 2.0 ** -8 * scale, (e_w, scale)                 # winograd: + fp16 roundings of the transformed inputs / weights
    # the operator with its two per-edge gate convolutions in Winograd form
    E, h, w = 6, 16, 64
    sd = deterministic_state_dict(_SD(oupd.empty_state_dict()), seed=7)
    net, inp, corr, flow = _update_inputs(E, h, w, seed=E + h)
    ii = torch.tensor([0, 0, 1, 2, 2, 2], dtype=torch.int64)
    inp = inp[[int(torch.nonzero(ii == f)[0]) for f in ii.tolist()]]
    with torch.no_grad():
        ref = oupd.update_forward(sd, net.half(), inp.half(), corr.half(), flow, ii, autocast=True)
    db.set_option("conv_wino", 1)
    try:
        mod = UpdateModule(share_inp_by_source_frame=True).load_state_dict(sd)
    finally:
        db.set_option("conv_wino", 0)
    assert mod.params["zr_e"].layout == LAYOUT_WINO and mod.params["q_e"].layout == LAYOUT_WINO
    n, d, wt, eta, up = mod(net[None].cuda().half(), inp[None].cuda().half(), corr[None].cuda().half(), flow[None].cuda(), ii.cuda(), None)
    torch.cuda.synchronize()
    _check_update((n[0], d[0], wt[0], eta[0], up[0]), ref, 3)
