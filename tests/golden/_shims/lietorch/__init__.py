"""CPU stand-in for the un-vendored ``lietorch`` (reference .gitmodules:1-3) -- GOLDEN GENERATION ONLY.

Used solely by tests/golden/make_golden.py to import the reference's Python
formulation (geom/ba.py, geom/projective_ops.py) on CPU.  Implements the SE3
subset listed in SURVEY.md Appendix C with plain torch ops; semantics follow the
device helpers of src/droid_kernels.cu:67-184, 886-904.  Not part of the product.
"""
import torch


def _cross(a, b):
    return torch.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                        a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def _rot(q, X):
    uv = 2.0 * _cross(q[..., :3], X)
    return X + q[..., 3:4] * uv + _cross(q[..., :3], uv)


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by + ay * bw + az * bx - ax * bz,
                        aw * bz + az * bw + ax * by - ay * bx,
                        aw * bw - ax * bx - ay * by - az * bz], -1)


class _Group:
    manifold_dim = 6
    embedded_dim = 7

    def __init__(self, data):
        self.data = data

    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    def __getitem__(self, index):
        return self.__class__(self.data[index])

    def detach(self):
        return self.__class__(self.data.detach())


class SE3(_Group):
    def inv(self):
        t, q = self.data[..., :3], self.data[..., 3:]
        qi = torch.cat([-q[..., :3], q[..., 3:]], -1)
        return SE3(torch.cat([-_rot(qi, t), qi], -1))

    def __mul__(self, other):
        t, q = self.data[..., :3], self.data[..., 3:]
        if isinstance(other, SE3):
            t2, q2 = other.data[..., :3], other.data[..., 3:]
            return SE3(torch.cat([t + _rot(q, t2), _qmul(q, q2)], -1))
        X = other                                   # homogeneous points [...,4]
        Y = _rot(q, X[..., :3]) + X[..., 3:4] * t
        return torch.cat([Y, X[..., 3:4]], -1)

    def adjT(self, X):
        t, q = self.data[..., :3], self.data[..., 3:]
        qi = torch.cat([-q[..., :3], q[..., 3:]], -1)
        a = _rot(qi, X[..., :3])
        b = _rot(qi, X[..., 3:]) + _rot(qi, _cross(X[..., :3], t.expand_as(X[..., :3])))
        return torch.cat([a, b], -1)

    @staticmethod
    def exp(xi):
        tau, phi = xi[..., :3], xi[..., 3:]
        th2 = (phi * phi).sum(-1, keepdim=True)
        th = th2.sqrt()
        small = th2 < 1e-8
        ths = torch.where(small, torch.ones_like(th), th)
        imag = torch.where(small, 0.5 - th2 / 48.0, torch.sin(0.5 * ths) / ths)
        real = torch.where(small, 1.0 - th2 / 8.0, torch.cos(0.5 * ths))
        q = torch.cat([imag * phi, real], -1)
        big = th > 1e-4
        a = (1 - torch.cos(ths)) / (ths * ths)
        b = (ths - torch.sin(ths)) / (ths ** 3)
        c1 = _cross(phi, tau)
        c2 = _cross(phi, c1)
        t = tau + torch.where(big, a * c1 + b * c2, torch.zeros_like(c1))
        return SE3(torch.cat([t, q], -1))

    def retr(self, xi):
        return SE3.exp(xi) * self

    @staticmethod
    def Identity(*shape, device="cpu", dtype=torch.float32):
        d = torch.zeros(*shape, 7, dtype=dtype)
        d[..., 6] = 1.0
        return SE3(d)

    def log(self):
        """inverse of exp (lietorch SE3.log: SO3 log of the unit quaternion, tau = V(phi)^-1 t); the same expressions as
        oracle/se3.py se3_log, in float64, returned in the data's dtype"""
        d = self.data.double()
        t, v, w = d[..., :3], d[..., 3:6], d[..., 6:7]
        n2 = (v * v).sum(-1, keepdim=True)
        n = n2.sqrt()
        small = n2 < 1e-20
        ns = torch.where(small, torch.ones_like(n), n)
        k = torch.where(small, 2.0 / w - (2.0 / 3.0) * n2 / w ** 3, 2.0 * torch.atan2(ns, w) / ns)
        k = torch.where((~small) & (w < 0), 2.0 * torch.atan(ns / torch.where(w == 0, torch.ones_like(w), w)) / ns, k)
        phi = k * v
        th2 = (phi * phi).sum(-1, keepdim=True)
        th = th2.sqrt()
        tiny = th < 1e-6
        ths = torch.where(tiny, torch.ones_like(th), th)
        c = torch.where(tiny, 1.0 / 12.0 + th2 / 720.0,
                        (1.0 - 0.5 * ths * torch.cos(0.5 * ths) / torch.sin(0.5 * ths)) / torch.where(tiny, torch.ones_like(th2), th2))
        c1 = _cross(phi, t)
        c2 = _cross(phi, c1)
        return torch.cat([t - 0.5 * c1 + c * c2, phi], -1).to(self.data.dtype)


class Sim3(_Group):
    manifold_dim = 7
    embedded_dim = 8


class SO3(_Group):
    manifold_dim = 3
    embedded_dim = 4


def cat(xs, dim=0):
    return xs[0].__class__(torch.cat([x.data for x in xs], dim))
