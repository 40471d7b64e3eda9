"""Oracle restatement of the reference's dense bundle adjustment ``ba_cuda``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows src/droid_kernels.cu of the reference:
  * per-edge residuals / Jacobians / Hessian blocks ... projective_transform_kernel :185-433
  * segment sums ..................................... accum_cuda / accum_kernel     :863-883, 957-1007
  * Schur blocks ..................................... schur_block, EEt6x6, Ev6x1     :1010-1102, 1231-1320
  * damped fp64 solve, failure -> zero update ......... SparseBlock::solve            :1201-1222
  * depth back-substitution incl. the row-skip quirk .. EvT6x1_kernel                 :1104-1124, 1417-1426
  * retractions ...................................... pose_retr / disp_retr kernels  :907-955
  * driver, depth prior, ordering of updates .......... ba_cuda                       :1323-1443
The index conventions are spelled out in SURVEY.md Appendix A.

All arithmetic runs in ``dtype`` (float64 = "truth", float32 = reference-like
rounding); the linear solve is always float64, like the reference's Eigen LLT.
"""
import numpy as np
import scipy.linalg

from . import se3

MIN_DEPTH = 0.25          # src/droid_kernels.cu:35
WEIGHT_SCALE = 0.001      # :314-315
ALPHA_DEPTH_PRIOR = 0.05  # :1405
STEREO_BASELINE = -0.1    # :230


def edge_terms(poses, disps, intrinsics, targets, weights, ii, jj, dtype=np.float64):
    """Per-edge, per-pixel quantities of projective_transform_kernel (:185-433).

    Shapes: poses [N,7]; disps [N,HW] (row-major ht x wd flattened) with ``wd`` given
    through ``targets`` [E,2,ht,wd]; returns a dict with
      r [E,2,HW], w [E,2,HW] (after MIN_DEPTH / 0.001 scaling, BEFORE stereo zeroing),
      wp [E,2,HW] (pose weights: stereo edges zeroed), Ji,Jj [E,2,6,HW], Jz [E,2,HW].
    """
    E, _, ht, wd = targets.shape
    HW = ht * wd
    fx, fy, cx, cy = [dtype(x) for x in intrinsics]
    ii = np.asarray(ii, dtype=np.int64)
    jj = np.asarray(jj, dtype=np.int64)
    poses = poses.astype(dtype)
    ti, qi = se3.pose_split(poses[ii])
    tj, qj = se3.pose_split(poses[jj])
    tij, qij = se3.se3_rel(ti, qi, tj, qj)
    stereo = ii == jj
    tij[stereo] = np.array([STEREO_BASELINE, 0, 0], dtype=dtype)
    qij[stereo] = np.array([0, 0, 0, 1], dtype=dtype)

    v, u = np.meshgrid(np.arange(ht, dtype=dtype), np.arange(wd, dtype=dtype), indexing="ij")
    X0 = np.stack([(u.reshape(-1) - cx) / fx, (v.reshape(-1) - cy) / fy,
                   np.ones(HW, dtype=dtype)], axis=-1)                          # [HW,3]
    h = disps.reshape(disps.shape[0], HW).astype(dtype)[ii]                      # [E,HW]
    X4 = np.concatenate([np.broadcast_to(X0, (E, HW, 3)), h[..., None]], axis=-1)
    Y = se3.se3_act(tij[:, None], qij[:, None], X4)                              # [E,HW,4]
    x, y, z = Y[..., 0], Y[..., 1], Y[..., 2]
    bad = z < MIN_DEPTH
    d = np.where(bad, dtype(0), dtype(1) / np.where(bad, dtype(1), z))
    d2 = d * d
    tg = targets.reshape(E, 2, HW).astype(dtype)
    wt = weights.reshape(E, 2, HW).astype(dtype)
    w = np.where(bad[:, None], dtype(0), dtype(WEIGHT_SCALE) * wt)
    r = np.stack([tg[:, 0] - (fx * d * x + cx), tg[:, 1] - (fy * d * y + cy)], axis=1)

    o = np.zeros_like(d)
    Jj_u = fx * np.stack([h * d, o, -x * h * d2, -x * y * d2, 1 + x * x * d2, -y * d], axis=1)
    Jj_v = fy * np.stack([o, h * d, -y * h * d2, -1 - y * y * d2, x * y * d2, x * d], axis=1)
    Jj = np.stack([Jj_u, Jj_v], axis=1)                                           # [E,2,6,HW]
    Jz = np.stack([fx * (tij[:, None, 0] * d - tij[:, None, 2] * (x * d2)),
                   fy * (tij[:, None, 1] * d - tij[:, None, 2] * (y * d2))], axis=1)  # [E,2,HW]
    # Ji = -Adj(Tij)^T Jj   (:334-335)
    Jjt = np.moveaxis(Jj, 2, -1)                                                  # [E,2,HW,6]
    Ji = -se3.se3_adjT(tij[:, None, None], qij[:, None, None], Jjt)
    Ji = np.moveaxis(Ji, -1, 2)
    wp = w.copy()
    wp[stereo] = 0                                                                # :332,365
    return dict(r=r, w=w, wp=wp, Ji=Ji, Jj=Jj, Jz=Jz, tij=tij, qij=qij)


def edge_blocks(T):
    """Hs[4,E,6,6] (ii,ij,ji,jj), vs[2,E,6], Eii,Eij[E,6,HW], Cii,bz[E,HW]  (:337-432)."""
    r, w, wp, Ji, Jj, Jz = T["r"], T["w"], T["wp"], T["Ji"], T["Jj"], T["Jz"]
    wJi = wp[:, :, None] * Ji
    wJj = wp[:, :, None] * Jj
    Hii = np.einsum("eckp,eclp->ekl", wJi, Ji)
    Hij = np.einsum("eckp,eclp->ekl", wJi, Jj)
    Hji = np.einsum("eckp,eclp->ekl", wJj, Ji)
    Hjj = np.einsum("eckp,eclp->ekl", wJj, Jj)
    vi = np.einsum("eckp,ecp->ek", wJi, r)
    vj = np.einsum("eckp,ecp->ek", wJj, r)
    Eii = np.einsum("eckp,ecp->ekp", wJi, Jz)
    Eij = np.einsum("eckp,ecp->ekp", wJj, Jz)
    Cii = np.sum(w * Jz * Jz, axis=1)
    bz = np.sum(w * r * Jz, axis=1)
    return np.stack([Hii, Hij, Hji, Hjj]), np.stack([vi, vj]), Eii, Eij, Cii, bz


def edge_stage(poses, disps, intrinsics, targets, weights, ii, jj, dtype=np.float64, chunk=None, pool=None):
    """edge_blocks(edge_terms(...)) evaluated in chunks of edges (bounded temporaries: the [E,2,6,HW] Jacobians of 4096
    edges would be 1.2 GB each) and, with a thread pool, on several host cores (numpy releases the GIL inside its
    kernels).  Per-edge arithmetic is untouched, so the result does not depend on `chunk` / `pool`."""
    E = len(ii)
    if chunk is None or chunk >= E:
        return edge_blocks(edge_terms(poses, disps, intrinsics, targets, weights, ii, jj, dtype=dtype))
    slices = [slice(a, min(E, a + chunk)) for a in range(0, E, chunk)]
    work = lambda sl: edge_blocks(edge_terms(poses, disps, intrinsics, targets[sl], weights[sl], ii[sl], jj[sl], dtype=dtype))
    parts = list(pool.map(work, slices)) if pool is not None else [work(sl) for sl in slices]
    return tuple(np.concatenate([p[k] for p in parts], axis=1 if k < 2 else 0) for k in range(6))


def _segsum(data, ix, jx):
    """accum_cuda: out[j] = sum_{i: ix[i]==jx[j]} data[i]   (:957-1007)."""
    out = np.zeros((len(jx),) + data.shape[1:], dtype=data.dtype)
    pos = {int(f): n for n, f in enumerate(jx)}
    for n, f in enumerate(ix):
        k = pos.get(int(f))
        if k is not None:
            out[k] += data[n]
    return out


def solve_damped(H, b, lm, ep):
    """SparseBlock::solve (:1201-1222): diag += ep + lm*diag, LLT in float64, zero on failure."""
    L = np.array(H, dtype=np.float64)
    dg = np.diag_indices_from(L)
    L[dg] += ep + lm * L[dg]
    # Eigen::SimplicialLLT reads the lower triangle only
    L = np.tril(L) + np.tril(L, -1).T
    try:
        if not np.all(np.isfinite(L)):
            raise np.linalg.LinAlgError("non-finite")
        c = scipy.linalg.cho_factor(L, lower=True, check_finite=False)
        x = scipy.linalg.cho_solve(c, np.asarray(b, dtype=np.float64), check_finite=False)
        ok = True
    except np.linalg.LinAlgError:
        x = np.zeros(len(b), dtype=np.float64)
        ok = False
    return x, ok


def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj,
       t0, t1, iterations, lm, ep, motion_only, dtype=np.float64, strict_q6=True,
       return_system=False, chunk=None, threads=1, alpha_map=None):
    """In-place Gauss-Newton on ``poses`` [buf,7] and ``disps`` [buf,ht,wd]  (ba_cuda :1323-1443).

    ``eta`` has one row per entry of ``unique(cat(arange(t0,t1), ii))``.
    Returns ``(dx [P,6], dz [K,HW])`` of the last iteration (``dz`` None when
    ``motion_only``), computed in ``dtype`` and stored back into the callers'
    arrays in their own dtype.  ``chunk`` / ``threads``: evaluate the per-edge stage in chunks of edges and the per-edge /
    per-depth-block stages on a thread pool (the multi-core CPU baseline of bench.py); results do not depend on them.  ``strict_q6`` reproduces EvT6x1_kernel's skip of
    rows whose relative pose index is <= 0 (:1114).
    """
    E, _, ht, wd = targets.shape
    HW = ht * wd
    ii = np.asarray(ii, dtype=np.int64)
    jj = np.asarray(jj, dtype=np.int64)
    P = t1 - t0
    ts = np.arange(t0, t1, dtype=np.int64)
    ii_exp = np.concatenate([ts, ii])
    jj_exp = np.concatenate([ts, jj])
    kx, kk_exp = np.unique(ii_exp, return_inverse=True)
    K = len(kx)
    dx = dz = None
    info = {}
    pool = None
    if threads and threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=threads)
        if chunk is None:
            chunk = max(8, -(-E // (4 * threads)))
    for _ in range(iterations):
        Hs, vs, Eii, Eij, Cii, bz = edge_stage(poses, disps, intrinsics, targets, weights, ii, jj, dtype=dtype, chunk=chunk, pool=pool)

        # pose x pose block A, rhs (update_lhs / update_rhs :1140-1182, called at :1385-1392)
        A = np.zeros((P, 6, P, 6), dtype=np.float64)
        bA = np.zeros((P, 6), dtype=np.float64)
        bi = np.concatenate([ii, ii, jj, jj]) - t0
        bj = np.concatenate([ii, jj, ii, jj]) - t0
        Hflat = Hs.reshape(-1, 6, 6).astype(np.float64)
        for n in range(len(bi)):
            if bi[n] >= 0 and bj[n] >= 0:
                A[bi[n], :, bj[n], :] += Hflat[n]
        vi = np.concatenate([ii, jj]) - t0
        vflat = vs.reshape(-1, 6).astype(np.float64)
        for n in range(len(vi)):
            if vi[n] >= 0:
                bA[vi[n]] += vflat[n]

        if motion_only:
            x, ok = solve_damped(A.reshape(6 * P, 6 * P), bA.reshape(-1), lm, ep)
            dx = x.reshape(P, 6).astype(np.float32).astype(dtype)
            _retract_poses(poses, dx, t0, t1, dtype)
            info = dict(ok=ok, H=A.reshape(6 * P, 6 * P), b=bA.reshape(-1)) if return_system else dict(ok=ok)
            continue

        # depth block with the sensor prior (:1405-1409)
        dsens = disps_sens.reshape(disps_sens.shape[0], HW).astype(dtype)[kx]
        dcur = disps.reshape(disps.shape[0], HW).astype(dtype)[kx]
        # the reference's constant 0.05 (:1405); alpha_map [buf,ht,wd] = the per-pixel extension (dh_ba_ex): zero
        # confidence switches the prior off at that pixel (eta damps it, like a pixel without sensor depth)
        alpha = dtype(ALPHA_DEPTH_PRIOR) if alpha_map is None else alpha_map.reshape(alpha_map.shape[0], HW).astype(dtype)[kx]
        m = ((dsens > 0) & (alpha > 0)).astype(dtype)
        C = _segsum(Cii, ii, kx) + m * alpha + (1 - m) * eta.reshape(-1, HW).astype(dtype)
        w = _segsum(bz, ii, kx) - m * alpha * (dcur - dsens)
        Q = dtype(1) / C

        Ei = _segsum(Eii, ii, ts)                                  # [P,6,HW] (:1411)
        Erow = np.concatenate([Ei, Eij], axis=0)                   # [P+E,6,HW] (:1412)

        # Schur complement (schur_block :1231-1320)
        S = np.zeros((P, 6, P, 6), dtype=np.float64)
        bS = np.zeros((P, 6), dtype=np.float64)
        rows_of = [[] for _ in range(K)]
        for n in range(len(jj_exp)):
            if t0 <= jj_exp[n] <= t1:                              # :1257 (j==t1 never occurs)
                rows_of[kk_exp[n]].append(n)
        def gram(k):
            rows = rows_of[k]
            if not rows:
                return None
            M = Erow[rows]                                         # [r,6,HW]
            return np.einsum("aip,p,bjp->aibj", M, Q[k], M)
        grams = list(pool.map(gram, range(K))) if pool is not None else [gram(k) for k in range(K)]
        for k in range(K):
            rows = rows_of[k]
            if not rows:
                continue
            G = grams[k]
            pk = jj_exp[rows] - t0
            for a in range(len(rows)):
                for b_ in range(len(rows)):
                    S[pk[a], :, pk[b_], :] += G[a, :, b_, :]
        # Ev6x1 over ALL rows, update_rhs drops negative pose indices (:1068-1102, 1317)
        qw = Q * w
        vrow = np.einsum("nip,np->ni", Erow, qw[kk_exp])
        for n in range(len(jj_exp)):
            p = jj_exp[n] - t0
            if p >= 0:
                bS[p] += vrow[n]

        Hsys = (A - S).reshape(6 * P, 6 * P)
        bsys = (bA - bS).reshape(-1)
        x, ok = solve_damped(Hsys, bsys, lm, ep)
        dx = x.reshape(P, 6).astype(np.float32).astype(dtype)      # reference casts dx to fp32 (:1213-1214)

        # back-substitution (:1417-1426)
        prel = jj_exp - t0
        use = (prel > 0) & (prel < P) if strict_q6 else (prel >= 0) & (prel < P)
        dw = np.zeros((len(jj_exp), HW), dtype=dtype)
        dw[use] = np.einsum("nip,ni->np", Erow[use], dx[prel[use]])
        dz = Q * (w - _segsum(dw, ii_exp, kx))

        _retract_poses(poses, dx, t0, t1, dtype)
        dflat = disps.reshape(disps.shape[0], HW)
        dflat[kx] = (dflat[kx].astype(dtype) + dz).astype(disps.dtype)
        if not np.shares_memory(dflat, disps):                     # caller's buffer is not C-contiguous: reshape copied
            disps[...] = dflat.reshape(disps.shape)
        info = dict(ok=ok, H=Hsys, b=bsys, C=C, w=w, kx=kx, A=A, S=S, bA=bA, bS=bS,
                    Hs=Hs, vs=vs, Erow=Erow, Q=Q, ii_exp=ii_exp, jj_exp=jj_exp, kk_exp=kk_exp) if return_system else dict(ok=ok)
    if pool is not None:
        pool.shutdown()
    if return_system:
        return dx, dz, info
    return dx, dz


def _retract_poses(poses, dx, t0, t1, dtype):
    t, q = se3.pose_split(poses[t0:t1].astype(dtype))
    t1_, q1_ = se3.se3_retr(dx.astype(dtype), t, q)
    poses[t0:t1] = se3.pose_join(t1_, q1_).astype(poses.dtype)


def reprojection_cost(poses, disps, intrinsics, targets, weights, ii, jj):
    """Weighted squared reprojection error (test helper; float64)."""
    T = edge_terms(poses, disps, intrinsics, targets, weights, ii, jj, dtype=np.float64)
    return float(np.sum(T["w"] * T["r"] ** 2))


# --------------------------------------------------------------------------------------
# dense "Python fallback" formulation (droid_slam/geom/ba.py:31-106 + geom/chol.py:46-73)
# used only to pin the oracle against the reference's own Python code (tests/golden)
# --------------------------------------------------------------------------------------
def ba_dense_python_formulation(poses, disps, intrinsics, targets, weights, eta, ii, jj,
                                fixedp, lm=1e-4, ep=0.1, min_depth=0.2, dtype=np.float64):
    """One iteration of geom/ba.py:BA semantics (differences to ba_cuda: Q8, Q9 of SURVEY.md 7).

    * validity uses ``Z > min_depth`` of BOTH the source point (always 1) and the
      transformed point, and the projection clamps ``Z < 0.5*min_depth -> 1``
      (projective_ops.py:47-82,185);
    * damping ``H += (ep + lm*H) * I`` is applied BEFORE the Schur complement
      (chol.py:54-62) and ``C += eta + 1e-7`` (ba.py:91);
    * all poses ``>= fixedp`` are free, all frames in ``unique(ii)`` carry depth;
    * disps > 10 are zeroed, then clamped at 0 (ba.py:102-103).
    Returns new (poses, disps, dx, dz) without modifying the inputs.
    """
    E, _, ht, wd = targets.shape
    HW = ht * wd
    N = poses.shape[0]
    fx, fy, cx, cy = [dtype(x) for x in intrinsics]
    ii = np.asarray(ii, dtype=np.int64)
    jj = np.asarray(jj, dtype=np.int64)
    posesd = poses.astype(dtype)
    ti, qi = se3.pose_split(posesd[ii])
    tj, qj = se3.pose_split(posesd[jj])
    tij, qij = se3.se3_rel(ti, qi, tj, qj)
    stereo = ii == jj
    tij[stereo] = np.array([STEREO_BASELINE, 0, 0], dtype=dtype)
    qij[stereo] = np.array([0, 0, 0, 1], dtype=dtype)
    v, u = np.meshgrid(np.arange(ht, dtype=dtype), np.arange(wd, dtype=dtype), indexing="ij")
    X0 = np.stack([(u.reshape(-1) - cx) / fx, (v.reshape(-1) - cy) / fy, np.ones(HW, dtype=dtype)], -1)
    h = disps.reshape(N, HW).astype(dtype)[ii]
    X4 = np.concatenate([np.broadcast_to(X0, (E, HW, 3)), h[..., None]], -1)
    Y = se3.se3_act(tij[:, None], qij[:, None], X4)
    X, Yc, Z, D = Y[..., 0], Y[..., 1], Y[..., 2], Y[..., 3]
    valid = (Z > min_depth).astype(dtype)                      # X0.z == 1 > min_depth always
    Zc = np.where(Z < 0.5 * min_depth, np.ones_like(Z), Z)
    d = 1.0 / Zc
    coords = np.stack([fx * X * d + cx, fy * Yc * d + cy], axis=1)     # [E,2,HW]
    o = np.zeros_like(d)
    # Jp [2,4] @ Ja [4,6]  (projective_ops.py:64-80, 93-122)
    Jp = np.stack([np.stack([fx * d, o, -fx * X * d * d, o], -1),
                   np.stack([o, fy * d, -fy * Yc * d * d, o], -1)], axis=-2)      # [E,HW,2,4]
    Ja = np.stack([np.stack([D, o, o, o, Z, -Yc], -1),
                   np.stack([o, D, o, -Z, o, X], -1),
                   np.stack([o, o, D, Yc, -X, o], -1),
                   np.stack([o, o, o, o, o, o], -1)], axis=-2)                    # [E,HW,4,6]
    Jj = Jp @ Ja                                                                  # [E,HW,2,6]
    Ji = -se3.se3_adjT(tij[:, None, None], qij[:, None, None], Jj)
    Jzp = se3.se3_act(tij[:, None], qij[:, None],
                      np.broadcast_to(np.array([0, 0, 0, 1], dtype=dtype), (E, HW, 4)))
    Jz = (Jp @ Jzp[..., None])[..., 0]                                            # [E,HW,2]

    r = np.moveaxis(targets.reshape(E, 2, HW).astype(dtype) - coords, 1, -1)      # [E,HW,2]
    w = dtype(0.001) * valid[..., None] * np.moveaxis(weights.reshape(E, 2, HW).astype(dtype), 1, -1)

    wJi = w[..., None] * Ji
    wJj = w[..., None] * Jj
    Hii = np.einsum("epck,epcl->ekl", wJi, Ji)
    Hij = np.einsum("epck,epcl->ekl", wJi, Jj)
    Hji = np.einsum("epck,epcl->ekl", wJj, Ji)
    Hjj = np.einsum("epck,epcl->ekl", wJj, Jj)
    vi = np.einsum("epck,epc->ek", wJi, r)
    vj = np.einsum("epck,epc->ek", wJj, r)
    Ei = np.einsum("epck,epc->ekp", wJi, Jz)
    Ej = np.einsum("epck,epc->ekp", wJj, Jz)
    wk = np.sum(w * r * Jz, axis=-1)
    Ck = np.sum(w * Jz * Jz, axis=-1)

    kx, kk = np.unique(ii, return_inverse=True)
    M = len(kx)
    Pn = N - fixedp
    pi, pj = ii - fixedp, jj - fixedp
    H = np.zeros((Pn, 6, Pn, 6), dtype=dtype)
    Em = np.zeros((Pn, 6, M, HW), dtype=dtype)
    vv = np.zeros((Pn, 6), dtype=dtype)
    for e in range(E):
        a, b = pi[e], pj[e]
        if a >= 0:
            H[a, :, a, :] += Hii[e]
            vv[a] += vi[e]
            Em[a, :, kk[e]] += Ei[e]
        if b >= 0:
            H[b, :, b, :] += Hjj[e]
            vv[b] += vj[e]
            Em[b, :, kk[e]] += Ej[e]
        if a >= 0 and b >= 0:
            H[a, :, b, :] += Hij[e]
            H[b, :, a, :] += Hji[e]
    C = np.zeros((M, HW), dtype=dtype)
    wv = np.zeros((M, HW), dtype=dtype)
    np.add.at(C, kk, Ck)
    np.add.at(wv, kk, wk)
    C = C + eta.reshape(M, HW).astype(dtype) + dtype(1e-7)

    Hm = H.reshape(6 * Pn, 6 * Pn).copy()
    dg = np.diag_indices_from(Hm)
    Hm[dg] += ep + lm * Hm[dg]
    Ef = Em.reshape(6 * Pn, M * HW)
    Qf = (1.0 / C).reshape(-1)
    S = Hm - (Ef * Qf) @ Ef.T
    vs_ = vv.reshape(-1) - Ef @ (Qf * wv.reshape(-1))
    try:
        c = scipy.linalg.cho_factor(S, lower=True)
        dxv = scipy.linalg.cho_solve(c, vs_)
    except np.linalg.LinAlgError:
        dxv = np.zeros_like(vs_)
    dzv = Qf * (wv.reshape(-1) - Ef.T @ dxv)
    dx = dxv.reshape(Pn, 6)
    dz = dzv.reshape(M, HW)

    new_poses = posesd.copy()
    t, q = se3.pose_split(posesd[fixedp:])
    t1_, q1_ = se3.se3_retr(dx, t, q)
    new_poses[fixedp:] = se3.pose_join(t1_, q1_)
    new_disps = disps.reshape(N, HW).astype(dtype).copy()
    new_disps[kx] += dz
    new_disps = np.where(new_disps > 10, 0.0, new_disps)
    new_disps = np.clip(new_disps, 0.0, None).reshape(disps.shape)
    return new_poses, new_disps, dx, dz
