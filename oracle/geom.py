"""Oracle restatement of the geometry kernels behind droid_backends and of the Python reprojection.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  projmap ............... projmap_kernel          src/droid_kernels.cu:436-525
  frame_distance ........ frame_distance_kernel   src/droid_kernels.cu:527-666
  depth_filter .......... depth_filter_kernel     src/droid_kernels.cu:670-784
  iproj ................. iproj_kernel            src/droid_kernels.cu:788-859
  projective_transform .. droid_slam/geom/projective_ops.py:165-198 (iproj :23-44, actp :85-162, proj :47-82)
"""
import numpy as np
from . import se3

MIN_DEPTH = 0.25      # kernels (src/droid_kernels.cu:35)
MIN_DEPTH_PY = 0.2    # Python reprojection (geom/projective_ops.py:6)


def _grid(ht, wd, intr, dtype):
    fx, fy, cx, cy = [dtype(x) for x in intr]
    v, u = np.meshgrid(np.arange(ht, dtype=dtype), np.arange(wd, dtype=dtype), indexing="ij")
    return u.reshape(-1), v.reshape(-1), (u.reshape(-1) - cx) / fx, (v.reshape(-1) - cy) / fy


def _transform(poses, disps, intr, ii, jj, dtype, stereo_override=False):
    N, ht, wd = disps.shape
    HW = ht * wd
    u, v, X, Y = _grid(ht, wd, intr, dtype)
    P = poses.astype(dtype)
    ti, qi = se3.pose_split(P[ii]); tj, qj = se3.pose_split(P[jj])
    tij, qij = se3.se3_rel(ti, qi, tj, qj)
    if stereo_override:
        st = np.asarray(ii) == np.asarray(jj)
        tij[st] = np.array([-0.1, 0, 0], dtype=dtype)
        qij[st] = np.array([0, 0, 0, 1], dtype=dtype)
    h = disps.reshape(N, HW).astype(dtype)[ii]
    X4 = np.stack([np.broadcast_to(X, h.shape), np.broadcast_to(Y, h.shape), np.ones_like(h), h], -1)
    Xj = se3.se3_act(tij[:, None], qij[:, None], X4)
    return u, v, X4, Xj, tij, qij


def projmap(poses, disps, intr, ii, jj, dtype=np.float64):
    """-> coords [M,ht,wd,3] (channel 2 stays 0), valid [M,ht,wd,1]."""
    N, ht, wd = disps.shape
    fx, fy, cx, cy = [dtype(x) for x in intr]
    u, v, X4, Xj, _, _ = _transform(poses, disps, intr, ii, jj, dtype)
    M = len(ii)
    z = Xj[..., 2]
    ok = z > 0.01
    zs = np.where(ok, z, 1.0)
    cu = np.where(ok, fx * (Xj[..., 0] / zs) + cx, u[None])
    cv = np.where(ok, fy * (Xj[..., 1] / zs) + cy, v[None])
    coords = np.stack([cu, cv, np.zeros_like(cu)], -1).reshape(M, ht, wd, 3)
    valid = (z > MIN_DEPTH).astype(dtype).reshape(M, ht, wd, 1)
    return coords, valid


def frame_distance(poses, disps, intr, ii, jj, beta, dtype=np.float64):
    """-> dist [M]: beta*(full motion flow) + (1-beta)*(translation-only flow), 1000 if < 75 % valid."""
    fx, fy, cx, cy = [dtype(x) for x in intr]
    u, v, X4, Xj, tij, qij = _transform(poses, disps, intr, ii, jj, dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        du = fx * (Xj[..., 0] / Xj[..., 2]) + cx - u
        dv = fy * (Xj[..., 1] / Xj[..., 2]) + cy - v
        d1 = np.sqrt(du * du + dv * dv)
        ok1 = Xj[..., 2] > MIN_DEPTH
        Xt = X4[..., :3] + X4[..., 3:4] * tij[:, None]
        du = fx * (Xt[..., 0] / Xt[..., 2]) + cx - u
        dv = fy * (Xt[..., 1] / Xt[..., 2]) + cy - v
        d2 = np.sqrt(du * du + dv * dv)
        ok2 = Xt[..., 2] > MIN_DEPTH
    HW = u.shape[0]
    accum = np.sum(np.where(ok1, beta * d1, 0.0), -1) + np.sum(np.where(ok2, (1 - beta) * d2, 0.0), -1)
    valid = np.sum(np.where(ok1, beta, 0.0), -1) + np.sum(np.where(ok2, 1 - beta, 0.0), -1)
    total = np.full(len(ii), HW * (beta + (1 - beta)), dtype=dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(valid / (total + 1e-8) < 0.75, 1000.0, accum / valid)


def iproj(poses, disps, intr, dtype=np.float64):
    """-> points [N,ht,wd,3] = (T_n * X)[:3] / disp   (NB: acts with the pose itself, not its inverse)."""
    N, ht, wd = disps.shape
    u, v, X, Y = _grid(ht, wd, intr, dtype)
    h = disps.reshape(N, -1).astype(dtype)
    X4 = np.stack([np.broadcast_to(X, h.shape), np.broadcast_to(Y, h.shape), np.ones_like(h), h], -1)
    t, q = se3.pose_split(poses[:N].astype(dtype))
    Xj = se3.se3_act(t[:, None], q[:, None], X4)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (Xj[..., :3] / Xj[..., 3:4]).reshape(N, ht, wd, 3)


def depth_filter(poses, disps, intr, ix, thresh, dtype=np.float64):
    """-> counter [M,ht,wd]: number of neighbours (ix-1,-2,-3,+3,+4,+5) with a consistent depth."""
    N, ht, wd = disps.shape
    fx, fy, cx, cy = [dtype(x) for x in intr]
    M = len(ix)
    counter = np.zeros((M, ht * wd), dtype=dtype)
    D = disps.astype(dtype)
    for m in range(M):
        i = int(ix[m])
        for nb in range(6):
            j = i - nb - 1 if nb < 3 else i + nb
            if j < 0 or j >= N:
                continue
            u, v, X4, Xj, _, _ = _transform(poses, disps, intr, np.array([i]), np.array([j]), dtype)
            Xj = Xj[0]
            with np.errstate(divide="ignore", invalid="ignore"):
                uj = fx * (Xj[:, 0] / Xj[:, 2]) + cx
                vj = fy * (Xj[:, 1] / Xj[:, 2]) + cy
                dj = Xj[:, 3] / Xj[:, 2]
            u0 = np.floor(uj); v0 = np.floor(vj)
            inb = np.isfinite(uj) & np.isfinite(vj) & (u0 >= 0) & (v0 >= 0) & (u0 < wd - 1) & (v0 < ht - 1)
            u0i = np.where(inb, u0, 0).astype(np.int64); v0i = np.where(inb, v0, 0).astype(np.int64)
            t = dtype(thresh[m])
            hit = np.zeros(ht * wd, dtype=bool)
            with np.errstate(divide="ignore", invalid="ignore"):
                for (a, b) in ((0, 0), (0, 1), (1, 0), (1, 1)):
                    dn = D[j, v0i + a, u0i + b]
                    hit |= np.abs(1.0 / dj - 1.0 / dn) < t
            counter[m] += (hit & inb).astype(dtype)
    return counter.reshape(M, ht, wd)


def projective_transform(poses, disps, intr, ii, jj, dtype=np.float64):
    """Python reprojection (projective_ops.py:165-198): coords [E,ht,wd,2], valid [E,ht,wd,1].

    Differences to the kernels: stereo override, Z < 0.1 -> 1, valid = Z > 0.2.
    ``intr`` is one 4-vector (the callers pass per-frame rows that are all equal)."""
    N, ht, wd = disps.shape
    fx, fy, cx, cy = [dtype(x) for x in intr]
    u, v, X4, Xj, _, _ = _transform(poses, disps, intr, ii, jj, dtype, stereo_override=True)
    E = len(ii)
    Z = Xj[..., 2]
    Zc = np.where(Z < 0.5 * MIN_DEPTH_PY, np.ones_like(Z), Z)
    d = 1.0 / Zc
    coords = np.stack([fx * (Xj[..., 0] * d) + cx, fy * (Xj[..., 1] * d) + cy], -1).reshape(E, ht, wd, 2)
    valid = ((Z > MIN_DEPTH_PY) & (X4[..., 2] > MIN_DEPTH_PY)).astype(dtype).reshape(E, ht, wd, 1)
    return coords, valid
