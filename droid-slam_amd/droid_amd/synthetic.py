"""Seeded synthetic frame graphs for the BASELINE.json configs.

Host-side only (numpy); used by tests/ and bench.py to build identical inputs for
the HIP path and for the CPU oracle.  No dataset or checkpoint is available, so
every input is synthetic: a bounded smooth camera trajectory, a smooth random
inverse-depth scene, directed edges (temporal neighbourhood + seeded loop
closures, optionally stereo self-edges), targets = ground-truth reprojection +
pixel noise, confidence weights in (0,1).

Pose convention (reference: src/droid_kernels.cu:105-116, 886-904):
world->camera ``[tx,ty,tz,qx,qy,qz,qw]``, left updates ``exp(xi)*T``.
"""
from dataclasses import dataclass
import numpy as np

HT, WD = 48, 64
INTRINSICS = (32.0, 32.0, 32.0, 24.0)   # TartanAir 320*0.8/8 (evaluation_scripts/test_tartanair.py:28,51)


@dataclass
class GraphConfig:
    name: str
    n_frames: int
    n_edges: int
    stereo: bool = False
    radius: int = 3          # temporal neighbourhood |i-j| <= radius
    lm: float = 1e-4
    ep: float = 0.1
    itrs: int = 2
    sensor_depth: bool = False


CONFIGS = {
    # BASELINE.json configs[0..4]
    "C1": GraphConfig("C1", 8, 32, radius=2),
    "C2": GraphConfig("C2", 64, 512),
    "C3": GraphConfig("C3", 512, 4096, lm=1e-5, ep=1e-2),
    "C4": GraphConfig("C4", 512, 4096, lm=1e-5, ep=1e-2),
    "C5": GraphConfig("C5", 1024, 8192, stereo=True, lm=1e-5, ep=1e-2, sensor_depth=True),
}


# ---------------------------------------------------------------- SE3 (numpy, float64)
def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def _rot(q, X):
    uv = 2.0 * _cross(q[..., :3], X)
    return X + q[..., 3:4] * uv + _cross(q[..., :3], uv)


def _qmul(a, b):
    ax, ay, az, aw = [a[..., i] for i in range(4)]
    bx, by, bz, bw = [b[..., i] for i in range(4)]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def se3_exp(xi):
    tau, phi = xi[..., :3], xi[..., 3:]
    th2 = np.sum(phi * phi, -1, keepdims=True)
    th = np.sqrt(th2)
    small = th < 1e-6
    ths = np.where(small, 1.0, th)
    imag = np.where(small, 0.5 - th2 / 48.0, np.sin(0.5 * ths) / ths)
    real = np.where(small, 1.0 - th2 / 8.0, np.cos(0.5 * ths))
    q = np.concatenate([imag * phi, real], -1)
    a = np.where(small, 0.5, (1 - np.cos(ths)) / (ths * ths))
    b = np.where(small, 1.0 / 6.0, (ths - np.sin(ths)) / (ths ** 3))
    c1 = _cross(phi, tau)
    t = tau + a * c1 + b * _cross(phi, c1)
    return np.concatenate([t, q], -1)


def se3_compose(A, B):
    """A * B on [...,7] pose arrays."""
    return np.concatenate([A[..., :3] + _rot(A[..., 3:], B[..., :3]), _qmul(A[..., 3:], B[..., 3:])], -1)


def reproject(poses, disps, intr, ii, jj):
    """Ground-truth reprojection ii->jj: coords [E,2,HW] and depth Z [E,HW] (float64)."""
    fx, fy, cx, cy = intr
    N, ht, wd = disps.shape
    HW = ht * wd
    v, u = np.meshgrid(np.arange(ht, dtype=np.float64), np.arange(wd, dtype=np.float64), indexing="ij")
    X0 = np.stack([(u.ravel() - cx) / fx, (v.ravel() - cy) / fy, np.ones(HW)], -1)
    Ti, Tj = poses[ii], poses[jj]
    qi_inv = np.concatenate([-Ti[:, 3:6], Ti[:, 6:7]], -1)
    qij = _qmul(Tj[:, 3:], qi_inv)
    tij = Tj[:, :3] - _rot(qij, Ti[:, :3])
    st = ii == jj
    tij[st] = [-0.1, 0, 0]
    qij[st] = [0, 0, 0, 1]
    h = disps.reshape(N, HW)[ii]
    Y = _rot(qij[:, None], np.broadcast_to(X0, (len(ii), HW, 3))) + h[..., None] * tij[:, None]
    Z = Y[..., 2]
    Zs = np.where(Z < 0.05, 1.0, Z)
    coords = np.stack([fx * Y[..., 0] / Zs + cx, fy * Y[..., 1] / Zs + cy], 1)
    return coords, Z


# ---------------------------------------------------------------- graph construction
def make_edges(cfg: GraphConfig, rng):
    N = cfg.n_frames
    es = []
    if cfg.stereo:
        es += [(i, i) for i in range(N)]
    for i in range(N):
        for j in range(N):
            if i != j and abs(i - j) <= cfg.radius:
                es.append((i, j))
    rest = cfg.n_edges - len(es)
    assert rest >= 0, "n_edges smaller than the temporal neighbourhood"
    if N <= 8:
        # C1: first pairs with |i-j| == radius+1 in lexicographic order
        extra = [(i, j) for i in range(N) for j in range(N) if abs(i - j) == cfg.radius + 1][:rest]
        es += extra
    else:
        have = set(es)
        while rest > 0:
            i, j = (int(x) for x in rng.integers(0, N, 2))
            if abs(i - j) <= cfg.radius or (i, j) in have:
                continue
            have.add((i, j)); have.add((j, i))
            es += [(i, j), (j, i)]
            rest -= 2
    es = es[:cfg.n_edges]
    ii = np.array([e[0] for e in es], dtype=np.int64)
    jj = np.array([e[1] for e in es], dtype=np.int64)
    return ii, jj


def _smooth_field(rng, n, ht, wd, lo, hi):
    """Smooth random field per frame: bilinear upsampling of a coarse grid."""
    gh, gw = 4, 5
    g = rng.uniform(lo, hi, (n, gh, gw))
    ys = np.linspace(0, gh - 1, ht)
    xs = np.linspace(0, gw - 1, wd)
    y0 = np.clip(np.floor(ys).astype(int), 0, gh - 2); fy = ys - y0
    x0 = np.clip(np.floor(xs).astype(int), 0, gw - 2); fx = xs - x0
    top = g[:, y0][:, :, x0] * (1 - fx) + g[:, y0][:, :, x0 + 1] * fx
    bot = g[:, y0 + 1][:, :, x0] * (1 - fx) + g[:, y0 + 1][:, :, x0 + 1] * fx
    return top * (1 - fy)[None, :, None] + bot * fy[None, :, None]


def depth_confidence(n_frames, ht=HT, wd=WD, seed=1234):
    """Seeded NON-constant per-pixel weight of the sensor-depth prior (BASELINE.json configs[4]: "per-pixel depth-confidence
    weights"): a smooth field per frame in [0.005, 0.25] -- the reference's constant is 0.05 (src/droid_kernels.cu:1405) -- with
    5 % of the pixels at zero confidence (the prior switched off there).  Its own random stream: the graphs of make_graph() and
    every golden made from them are unchanged."""
    rng = np.random.default_rng(seed + 7919)
    conf = _smooth_field(rng, n_frames, ht, wd, 0.005, 0.25)
    conf *= (rng.uniform(size=conf.shape) > 0.05)
    return conf.astype(np.float32)


def make_graph(cfg, seed=1234, ht=HT, wd=WD, with_features=False, feature_dim=128):
    """Build one synthetic BA problem.  Returns a dict of numpy arrays (float32 / int64)."""
    if isinstance(cfg, str):
        cfg = CONFIGS[cfg]
    rng = np.random.default_rng(seed)
    N = cfg.n_frames
    n = np.arange(N, dtype=np.float64)[:, None]
    amp = np.array([0.30, 0.20, 0.15, 0.06, 0.08, 0.05])
    freq = np.array([1 / 37.0, 1 / 53.0, 1 / 71.0, 1 / 43.0, 1 / 61.0, 1 / 29.0])
    phase = rng.uniform(0, 2 * np.pi, 6)
    xi = amp * np.sin(2 * np.pi * freq * n + phase) - amp * np.sin(phase) + rng.normal(0, 0.01, (N, 6))
    xi[0] = 0
    poses_gt = se3_exp(xi)
    pert = rng.normal(0, 0.02, (N, 6)); pert[0] = 0
    poses0 = se3_compose(se3_exp(pert), poses_gt)
    disps_gt = _smooth_field(rng, N, ht, wd, 0.3, 2.0)
    disps0 = np.ones_like(disps_gt)

    ii, jj = make_edges(cfg, rng)
    E = len(ii)
    HW = ht * wd
    coords, Z = reproject(poses_gt, disps_gt, INTRINSICS, ii, jj)
    targets = coords + rng.normal(0, 0.5, coords.shape)
    weights = 1.0 / (1.0 + np.exp(-rng.normal(0, 1, (E, 2, HW))))
    weights *= (rng.uniform(size=(E, 1, HW)) > 0.05)
    # out-of-view / behind-camera targets carry no information
    inview = (Z > 0.3) & (coords[:, 0] > -8) & (coords[:, 0] < wd + 8) & (coords[:, 1] > -8) & (coords[:, 1] < ht + 8)
    weights *= inview[:, None]

    kx = np.unique(np.concatenate([np.arange(1, N), ii]))
    eta = 0.2 * rng.uniform(1e-6, 1e-3, (len(kx), ht, wd)) + 1e-7
    if cfg.sensor_depth:
        sens = disps_gt * (1 + rng.normal(0, 0.02, disps_gt.shape))
        sens *= (rng.uniform(size=disps_gt.shape) > 0.3)
    else:
        sens = np.zeros_like(disps_gt)

    out = dict(
        name=cfg.name, n_frames=N, ht=ht, wd=wd, t0=1, t1=N, lm=cfg.lm, ep=cfg.ep, itrs=cfg.itrs,
        intrinsics=np.array(INTRINSICS, dtype=np.float32),
        poses_gt=poses_gt.astype(np.float32), poses=poses0.astype(np.float32),
        disps_gt=disps_gt.astype(np.float32), disps=disps0.astype(np.float32),
        disps_sens=sens.astype(np.float32),
        ii=ii, jj=jj,
        targets=targets.reshape(E, 2, ht, wd).astype(np.float32),
        weights=weights.reshape(E, 2, ht, wd).astype(np.float32),
        eta=eta.astype(np.float32),
    )
    if cfg.sensor_depth:
        out["disps_conf"] = depth_confidence(N, ht, wd, seed)
    if with_features:
        rig = 2 if cfg.stereo else 1
        out["fmaps"] = rng.standard_normal((N, rig, feature_dim, ht, wd), dtype=np.float32).astype(np.float16)
        out["nets"] = np.tanh(rng.standard_normal((N, 128, ht, wd), dtype=np.float32)).astype(np.float16)
        out["inps"] = np.maximum(rng.standard_normal((N, 128, ht, wd), dtype=np.float32), 0).astype(np.float16)
    return out


def small_graph(n_frames=5, n_edges=None, seed=7, ht=12, wd=16, stereo=False, sensor_depth=False, radius=2):
    """Tiny problems for fast oracle tests (the oracle loops in Python)."""
    es = n_edges
    if es is None:
        es = sum(1 for i in range(n_frames) for j in range(n_frames) if i != j and abs(i - j) <= radius)
        es += n_frames if stereo else 0
    cfg = GraphConfig("tiny", n_frames, es, stereo=stereo, radius=radius, sensor_depth=sensor_depth)
    g = make_graph(cfg, seed=seed, ht=ht, wd=wd)
    # intrinsics scaled to the smaller image
    s = ht / HT
    g["intrinsics"] = (np.array(INTRINSICS) * s).astype(np.float32)
    # regenerate targets with the scaled intrinsics
    rng = np.random.default_rng(seed + 1)
    coords, Z = reproject(g["poses_gt"].astype(np.float64), g["disps_gt"].astype(np.float64),
                          tuple(g["intrinsics"].astype(np.float64)), g["ii"], g["jj"])
    E = len(g["ii"])
    g["targets"] = (coords + rng.normal(0, 0.1, coords.shape)).reshape(E, 2, ht, wd).astype(np.float32)
    w = 1.0 / (1.0 + np.exp(-rng.normal(0, 1, (E, 2, ht * wd))))
    w *= (Z > 0.3)[:, None]
    g["weights"] = w.reshape(E, 2, ht, wd).astype(np.float32)
    return g
