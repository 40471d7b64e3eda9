"""SE(3) helpers of the oracle (numpy, any float dtype, broadcasting over leading dims).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Pose layout everywhere: ``[tx, ty, tz, qx, qy, qz, qw]`` (world->camera), tangent
vectors ``xi = (tau, phi)``; updates are LEFT multiplications ``exp(xi) * T``.
Each function restates a device helper of the reference:

    so3_act   <- actSO3   src/droid_kernels.cu:67-77
    se3_act   <- actSE3   src/droid_kernels.cu:79-86
    se3_adjT  <- adjSE3   src/droid_kernels.cu:88-103
    se3_rel   <- relSE3   src/droid_kernels.cu:105-116
    so3_exp   <- expSO3   src/droid_kernels.cu:119-141
    se3_exp   <- expSE3   src/droid_kernels.cu:156-184
    se3_retr  <- retrSE3  src/droid_kernels.cu:886-904
and the lietorch group ops the Python callers use (projective_ops.py:174-193):
    se3_inv, se3_mul  (definitions in SURVEY.md Appendix C).
"""
import numpy as np


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def so3_act(q, X):
    """Rotate 3-vectors X by unit quaternions q=(x,y,z,w)."""
    qv = q[..., :3]
    uv = 2.0 * _cross(qv, X)
    return X + q[..., 3:4] * uv + _cross(qv, uv)


def se3_act(t, q, X4):
    """Homogeneous action  Y3 = R X3 + t X4,  Y4 = X4."""
    Y3 = so3_act(q, X4[..., :3]) + X4[..., 3:4] * t
    return np.concatenate([Y3, X4[..., 3:4]], axis=-1)


def quat_conj(q):
    return np.concatenate([-q[..., :3], q[..., 3:4]], axis=-1)


def quat_mul(a, b):
    """Hamilton product a*b, (x,y,z,w) layout (cf. retrSE3 / relSE3)."""
    ax, ay, az, aw = [a[..., i] for i in range(4)]
    bx, by, bz, bw = [b[..., i] for i in range(4)]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def se3_adjT(t, q, X6):
    """Dual adjoint  Y = Adj(T)^T X  on 6-vectors (tau, phi)."""
    qi = quat_conj(q)
    a = so3_act(qi, X6[..., :3])
    b = so3_act(qi, X6[..., 3:])
    u = _cross(X6[..., :3], t)           # = t x X  with the reference's sign convention
    # reference: u = (t2*X1 - t1*X2, t0*X2 - t2*X0, t1*X0 - t0*X1) = X[:3] x t
    b = b + so3_act(qi, u)
    return np.concatenate([a, b], axis=-1)


def se3_rel(ti, qi, tj, qj):
    """Relative transform  Tij = Tj * Ti^-1  -> (tij, qij)."""
    qij = quat_mul(qj, quat_conj(qi))
    tij = tj - so3_act(qij, ti)
    return tij, qij


def se3_inv(t, q):
    qi = quat_conj(q)
    return -so3_act(qi, t), qi


def se3_mul(t1, q1, t2, q2):
    return t1 + so3_act(q1, t2), quat_mul(q1, q2)


def so3_exp(phi):
    theta_sq = np.sum(phi * phi, axis=-1, keepdims=True)
    theta_p4 = theta_sq * theta_sq
    theta = np.sqrt(theta_sq)
    small = theta_sq < 1e-8
    safe = np.where(small, np.ones_like(theta), theta)
    imag = np.where(small,
                    0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_p4,
                    np.sin(0.5 * safe) / safe)
    real = np.where(small,
                    1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_p4,
                    np.cos(0.5 * safe))
    return np.concatenate([imag * phi, real], axis=-1)


def se3_exp(xi):
    """exp: (tau, phi) -> (t, q);  t = V(phi) tau with the 1e-4 threshold of expSE3."""
    tau, phi = xi[..., :3], xi[..., 3:]
    q = so3_exp(phi)
    theta_sq = np.sum(phi * phi, axis=-1, keepdims=True)
    theta = np.sqrt(theta_sq)
    big = theta > 1e-4
    safe_sq = np.where(big, theta_sq, np.ones_like(theta_sq))
    safe = np.where(big, theta, np.ones_like(theta))
    a = (1.0 - np.cos(safe)) / safe_sq
    b = (safe - np.sin(safe)) / (safe * safe_sq)
    c1 = _cross(phi, tau)
    c2 = _cross(phi, c1)
    t = tau + np.where(big, a * c1 + b * c2, np.zeros_like(c1))
    return t, q


def se3_log(t, q):
    """log: (t, q) -> xi = (tau, phi), the inverse of se3_exp (lietorch SE3.log, un-vendored: SO3 log of the unit
    quaternion, tau = V(phi)^-1 t).  Used by the callers of the path (droid_frontend.py:59-63, trajectory_filler.py:55-65)."""
    v, w = q[..., :3], q[..., 3:4]
    n2 = np.sum(v * v, -1, keepdims=True)
    n = np.sqrt(n2)
    small = n2 < 1e-20
    ns = np.where(small, 1.0, n)
    k = np.where(small, 2.0 / w - (2.0 / 3.0) * n2 / w ** 3, 2.0 * np.arctan2(ns, w) / ns)
    k = np.where((~small) & (w < 0), 2.0 * np.arctan(ns / np.where(w == 0, 1.0, w)) / ns, k)     # atan branch of the reference (no +pi unwrap)
    phi = k * v
    th2 = np.sum(phi * phi, -1, keepdims=True)
    th = np.sqrt(th2)
    tiny = th < 1e-6
    ths = np.where(tiny, 1.0, th)
    c = np.where(tiny, 1.0 / 12.0 + th2 / 720.0, (1.0 - 0.5 * ths * np.cos(0.5 * ths) / np.sin(0.5 * ths)) / np.where(tiny, 1.0, th2))
    c1 = _cross(phi, t)
    c2 = _cross(phi, c1)
    tau = t - 0.5 * c1 + c * c2
    return np.concatenate([tau, phi], -1)


def se3_retr(xi, t, q):
    """Left retraction  T <- exp(xi) * T."""
    dt, dq = se3_exp(xi)
    q1 = quat_mul(dq, q)
    t1 = so3_act(dq, t) + dt
    return t1, q1


def pose_split(poses):
    return poses[..., :3], poses[..., 3:7]


def pose_join(t, q):
    return np.concatenate([t, q], axis=-1)


def random_se3(rng, n, trans=1.0, rot=0.5, dtype=np.float64):
    """Seeded random poses (helper for tests)."""
    xi = np.concatenate([rng.normal(0, trans, (n, 3)), rng.normal(0, rot, (n, 3))], -1).astype(dtype)
    t, q = se3_exp(xi)
    return pose_join(t, q).astype(dtype)
