"""`lietorch` drop-in for the SE(3) subset that sits on DROID-SLAM's BA update path.

The reference imports the un-vendored princeton-vl/lietorch (reference .gitmodules:1-3) in depth_video.py:3,174,
factor_graph.py:2,6 and geom/projective_ops.py:4,94,174-193 and uses: SE3(data), .data, indexing / broadcasting views,
inv(), group product, action on homogeneous 4-vectors, adjT(), retr()/exp().  This package provides exactly that
surface on top of the HIP kernels of libdroid_hip (dh_se3_inv/mul/exp/retr/act4/adjT, semantics of the reference's
device helpers src/droid_kernels.cu:67-184,886-904); operands are broadcast to a common batch shape and sent to the
kernels as flat [n,7] arrays.  Inference only (no autograd); tensors must live on the ROCm device.
log / matrix serve the callers of the path (frontend motion model, trajectory filler, viewers); Sim3 / SO3 are used by
training and trajectory alignment only and raise NotImplementedError.
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


class Sim3(_Group):
    manifold_dim = 7
    embedded_dim = 8

    def __init__(self, data):
        super().__init__(data)

    def __mul__(self, other):
        raise NotImplementedError("Sim3 is used by training / evaluation only (SURVEY.md Appendix C)")


class SO3(_Group):
    manifold_dim = 3
    embedded_dim = 4


def cat(xs, dim=0):
    return xs[0].__class__(torch.cat([x.data for x in xs], dim if dim >= 0 else dim - 1))


def stack(xs, dim=0):
    return xs[0].__class__(torch.stack([x.data for x in xs], dim if dim >= 0 else dim - 1))
