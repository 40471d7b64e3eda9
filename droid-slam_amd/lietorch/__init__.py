"""`lietorch` drop-in for the SE(3) subset that sits on DROID-SLAM's BA update path.

The reference imports the un-vendored princeton-vl/lietorch (reference .gitmodules:1-3) in depth_video.py:3,174,
factor_graph.py:2,6 and geom/projective_ops.py:4,94,174-193 and uses: SE3(data), .data, indexing / broadcasting views,
inv(), group product, action on homogeneous 4-vectors, adjT(), retr()/exp().  This package provides exactly that
surface on top of the HIP kernels of libdroid_hip (dh_se3_inv/mul/exp/retr/act4/adjT, semantics of the reference's
device helpers src/droid_kernels.cu:67-184,886-904); operands are broadcast to a common batch shape and sent to the
kernels as flat [n,7] arrays.  Inference only (no autograd); tensors must live on the ROCm device.
log / matrix serve the callers of the path (frontend motion model, trajectory filler, viewers).  Sim3 (training and trajectory
alignment only) has its group algebra -- product, inverse, action, matrix -- as plain tensor arithmetic; Sim3.exp / log and
SO3 raise NotImplementedError.
"""
import torch

import droid_backends as _db


def _flat(t, last):
    return t.reshape(-1, last).contiguous().float()


class _Group:
    manifold_dim = 6
    embedded_dim = 7

    def __init__(self, data):
        self.data = data

    # -- tensor-like plumbing used by the reference ---------------------------------------------------------
    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    @property
    def dtype(self):
        return self.data.dtype

    def __getitem__(self, index):
        if not isinstance(index, tuple):
            index = (index,)
        return self.__class__(self.data[index + (slice(None),)] if Ellipsis not in index else self.data[index])

    def view(self, *dims):
        return self.__class__(self.data.view(*dims, self.embedded_dim))

    def detach(self):
        return self.__class__(self.data.detach())

    def to(self, *a, **k):
        return self.__class__(self.data.to(*a, **k))

    def cpu(self):
        return self.__class__(self.data.cpu())

    def vec(self):
        return self.data

    def __repr__(self):
        return "%s(%r)" % (self.__class__.__name__, self.data)


class SE3(_Group):
    """data [...,7] = (tx,ty,tz, qx,qy,qz,qw), world->camera in DROID-SLAM."""

    @staticmethod
    def Identity(*shape, device="cuda", dtype=torch.float32):
        d = torch.zeros(*shape, 7, device=device, dtype=dtype)
        d[..., 6] = 1.0
        return SE3(d)

    @staticmethod
    def IdentityLike(G):
        return SE3.Identity(*G.shape, device=G.device, dtype=G.dtype)

    def inv(self):
        out = _db.se3_op("inv", _flat(self.data, 7), _flat(self.data, 7))
        return SE3(out.view(self.data.shape).to(self.data.dtype))

    def __mul__(self, other):
        if isinstance(other, SE3):
            shape = torch.broadcast_shapes(self.data.shape, other.data.shape)
            a = _flat(self.data.expand(shape), 7); b = _flat(other.data.expand(shape), 7)
            return SE3(_db.se3_op("mul", a, b).view(shape).to(self.data.dtype))
        # action on homogeneous points [...,4] (projective_ops.py:87)
        X = other
        assert X.shape[-1] == 4, "SE3 acts on homogeneous 4-vectors"
        bshape = torch.broadcast_shapes(self.data.shape[:-1], X.shape[:-1])
        g = _flat(self.data.expand(bshape + (7,)), 7)
        x = X.expand(bshape + (4,)).reshape(-1, 1, 4).contiguous().float()
        return _db.se3_map("act4", g, x).view(bshape + (4,)).to(X.dtype)

    def act(self, X):
        return self * X

    def adjT(self, X):
        bshape = torch.broadcast_shapes(self.data.shape[:-1], X.shape[:-1])
        g = _flat(self.data.expand(bshape + (7,)), 7)
        x = X.expand(bshape + (6,)).reshape(-1, 1, 6).contiguous().float()
        return _db.se3_map("adjT", g, x).view(bshape + (6,)).to(X.dtype)

    @staticmethod
    def exp(xi):
        return SE3(_db.se3_op("exp", _flat(xi, 6), _flat(xi, 6)).view(xi.shape[:-1] + (7,)).to(xi.dtype))

    def retr(self, xi):
        shape = torch.broadcast_shapes(self.data.shape[:-1], xi.shape[:-1])
        a = _flat(xi.expand(shape + (6,)), 6); b = _flat(self.data.expand(shape + (7,)), 7)
        return SE3(_db.se3_op("retr", a, b).view(shape + (7,)).to(self.data.dtype))

    def log(self):
        """[...,7] -> tangent vectors [...,6] = (tau, phi) (droid_frontend.py:61, trajectory_filler.py:63)"""
        out = _db.se3_op("log", _flat(self.data, 7), _flat(self.data, 7))
        return out.view(self.data.shape[:-1] + (6,)).to(self.data.dtype)

    def matrix(self):
        """[...,7] -> homogeneous 4x4 matrices (viewers, visualization.py:86); plain tensor arithmetic, not on the hot path"""
        t, q = self.data[..., :3], self.data[..., 3:]
        x, y, z, w = q.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).view(q.shape[:-1] + (3, 3))
        M = torch.zeros(q.shape[:-1] + (4, 4), dtype=self.data.dtype, device=self.data.device)
        M[..., :3, :3] = R; M[..., :3, 3] = t; M[..., 3, 3] = 1.0
        return M

    def scale(self, s):
        """translation scaled by s (DepthVideo.normalize-style rescaling of a trajectory)"""
        s = torch.as_tensor(s, dtype=self.data.dtype, device=self.data.device)
        return SE3(torch.cat([self.data[..., :3] * s[..., None] if s.dim() else self.data[..., :3] * s, self.data[..., 3:]], -1))


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1); bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], -1)


def _qrot(q, v):
    u, w = q[..., :3], q[..., 3:]
    t = 2.0 * torch.cross(u, v, dim=-1)
    return v + w * t + torch.cross(u, t, dim=-1)


class Sim3(_Group):
    """similarity transforms x -> s R(q) x + t, data [...,8] = (t, q = xyzw, s).  Not on the BA update path (the reference
    constructs Sim3 only in training, geom/losses.py / train.py, and in the Sim3 branch of projective_ops.actp): the group
    algebra below is plain tensor arithmetic; exp / log (training only) are not provided."""
    manifold_dim = 7
    embedded_dim = 8

    def __init__(self, data):
        super().__init__(data.data if isinstance(data, _Group) and data.data.shape[-1] == 8 else
                         (torch.cat([data.data, torch.ones_like(data.data[..., :1])], -1) if isinstance(data, _Group) else data))

    @staticmethod
    def Identity(*shape, device="cuda", dtype=torch.float32):
        d = torch.zeros(*shape, 8, device=device, dtype=dtype)
        d[..., 6] = 1.0; d[..., 7] = 1.0
        return Sim3(d)

    def inv(self):
        t, q, s = self.data[..., :3], self.data[..., 3:7], self.data[..., 7:]
        qi = torch.cat([-q[..., :3], q[..., 3:]], -1)
        return Sim3(torch.cat([-_qrot(qi, t) / s, qi, 1.0 / s], -1))

    def __mul__(self, other):
        a, b = torch.broadcast_tensors(self.data, other.data)
        t = a[..., 7:] * _qrot(a[..., 3:7], b[..., :3]) + a[..., :3]
        return Sim3(torch.cat([t, _qmul(a[..., 3:7], b[..., 3:7]), a[..., 7:] * b[..., 7:]], -1))

    def act(self, X):
        """homogeneous points [...,4] (projective_ops.py:94 convention: translation scaled by the 4th coordinate)"""
        d = self.data
        while d.dim() < X.dim():
            d = d[..., None, :]
        p = d[..., 7:] * _qrot(d[..., 3:7].expand(X.shape[:-1] + (4,)), X[..., :3]) + d[..., :3] * X[..., 3:]
        return torch.cat([p, X[..., 3:]], -1)

    def matrix(self):
        M = SE3(self.data[..., :7]).matrix()
        M[..., :3, :3] = M[..., :3, :3] * self.data[..., 7:, None]
        return M

    @staticmethod
    def exp(xi):
        raise NotImplementedError("Sim3.exp is used by training only (SURVEY.md Appendix C)")

    def log(self):
        raise NotImplementedError("Sim3.log is used by training / evaluation only (SURVEY.md Appendix C)")


class SO3(_Group):
    manifold_dim = 3
    embedded_dim = 4


def cat(xs, dim=0):
    return xs[0].__class__(torch.cat([x.data for x in xs], dim if dim >= 0 else dim - 1))


def stack(xs, dim=0):
    return xs[0].__class__(torch.stack([x.data for x in xs], dim if dim >= 0 else dim - 1))
