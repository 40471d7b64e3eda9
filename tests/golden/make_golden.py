#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE's own Python code on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference CUDA extension cannot be built here (no nvcc / Eigen / NVIDIA GPU) and
its tests directory does not exist, so the only reference-produced vectors
available are those of its Python formulation of the same path:

  * geom/projective_ops.py:projective_transform  (reprojection + analytic Jacobians)
  * geom/ba.py:BA + geom/chol.py:schur_solve     (dense Gauss-Newton step, "Python fallback")
  * modules/corr.py:CorrBlock.corr + pyramid     (all-pairs volume + 3 avg-pools)
  * droid_net.py:UpdateModule, cvx_upsample      (ConvGRU update block)

The un-vendored dependencies are replaced by the stand-ins in tests/golden/_shims
(lietorch SE3 subset, torch_scatter, an empty droid_backends); the hard-coded
device="cuda" in projective_ops.py:177 is neutralised by giving that module a torch
proxy whose as_tensor ignores the device (the reference files are not modified).
Outputs: tests/golden/*.npz (committed).  tests/test_oracle_golden.py checks oracle/
against them; the -m gpu tests check the HIP path against the same files.
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/droid_slam"
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "droid-slam_amd"))

import lietorch                                   # shim
import geom.projective_ops as pops                # reference
from geom.ba import BA                            # reference
from modules.corr import CorrBlock                # reference
import droid_net as ref_net                       # reference
from droid_amd import synthetic as syn
from droid_amd.weights import fill_deterministic


class _SoftplusF32(torch.nn.Module):
    """torch.autocast on a CUDA/ROCm device runs softplus in float32 (it is on autocast's fp32 list); CPU autocast would
    run it in fp16.  GraphAgg's eta head (droid_net.py:53-56) is the one place of the update operator where the two
    policies differ, so the golden run casts like the GPU does."""

    def forward(self, x):
        return torch.nn.functional.softplus(x.float())


class _TorchProxy:
    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def as_tensor(*a, **kw):
        kw.pop("device", None)
        if not isinstance(a[0], torch.Tensor):
            kw.setdefault("dtype", torch.get_default_dtype())   # the literal [-0.1,0,...] follows the run's dtype
        return torch.as_tensor(*a, **kw)


pops.torch = _TorchProxy()


def golden_ba():
    torch.set_default_dtype(torch.float64)
    try:
        _golden_ba()
    finally:
        torch.set_default_dtype(torch.float32)


def _golden_ba():
    g = syn.small_graph(n_frames=5, seed=11, ht=12, wd=16, stereo=True)
    N, ht, wd = g["disps"].shape
    ii = torch.as_tensor(g["ii"]); jj = torch.as_tensor(g["jj"])
    f64 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64))
    poses = lietorch.SE3(f64(g["poses"])[None])
    disps = f64(g["disps"] * 0.9 + 0.05 * g["disps_gt"])[None]
    intr = f64(g["intrinsics"])[None, None].repeat(1, N, 1)
    E = len(g["ii"])
    target = f64(g["targets"]).permute(0, 2, 3, 1)[None].contiguous()      # [1,E,ht,wd,2]
    weight = f64(g["weights"]).permute(0, 2, 3, 1)[None].contiguous()
    kx = torch.unique(ii)
    rng = np.random.default_rng(5)
    eta = f64(0.2 * rng.uniform(1e-4, 1e-2, (1, len(kx), ht, wd)) + 1e-7)

    coords, valid, (Ji, Jj, Jz) = pops.projective_transform(poses, disps, intr, ii, jj, jacobian=True)
    poses1, disps1 = BA(target, weight, eta, poses, disps, intr, ii, jj, fixedp=1)
    poses2, disps2 = BA(target, weight, eta, poses1, disps1, intr, ii, jj, fixedp=1)
    np.savez_compressed(
        os.path.join(HERE, "ba_python.npz"),
        poses=poses.data[0].numpy(), disps=disps[0].numpy(), intrinsics=g["intrinsics"].astype(np.float64),
        targets=g["targets"].astype(np.float64), weights=g["weights"].astype(np.float64),
        eta=eta[0].numpy(), ii=g["ii"], jj=g["jj"], fixedp=1,
        coords=coords[0].numpy(), valid=valid[0].numpy(),
        Ji=Ji[0].numpy(), Jj=Jj[0].numpy(), Jz=Jz[0].numpy(),
        poses1=poses1.data[0].numpy(), disps1=disps1[0].numpy(),
        poses2=poses2.data[0].numpy(), disps2=disps2[0].numpy())
    print("ba_python: E=%d  |dpose|=%.3e  |ddisp|=%.3e" % (
        E, (poses1.data - poses.data).abs().max(), (disps1 - disps).abs().max()))


def golden_corr():
    rng = np.random.default_rng(21)
    E, C, h, w = 2, 16, 16, 16
    f1 = torch.as_tensor(rng.standard_normal((1, E, C, h, w)).astype(np.float32))
    f2 = torch.as_tensor(rng.standard_normal((1, E, C, h, w)).astype(np.float32))
    blk = CorrBlock(f1, f2, num_levels=4, radius=3)
    out = {"fmap1": f1[0].numpy(), "fmap2": f2[0].numpy()}
    for l, v in enumerate(blk.corr_pyramid):
        out["level%d" % l] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "corr_python.npz"), **out)
    print("corr_python:", [tuple(v.shape) for v in blk.corr_pyramid])


def golden_update():
    torch.manual_seed(0)
    m = ref_net.UpdateModule()
    fill_deterministic(m, seed=1234)
    m.eval()
    rng = np.random.default_rng(31)
    E, ht, wd = 4, 8, 12
    ii = torch.as_tensor([0, 0, 2, 2]); jj = torch.as_tensor([1, 2, 0, 1])
    net = torch.as_tensor(np.tanh(rng.standard_normal((1, E, 128, ht, wd))).astype(np.float32))
    inp = torch.as_tensor(np.maximum(rng.standard_normal((1, E, 128, ht, wd)), 0).astype(np.float32))
    corr = torch.as_tensor(rng.standard_normal((1, E, 196, ht, wd)).astype(np.float32))
    flow = torch.as_tensor((4 * rng.standard_normal((1, E, 4, ht, wd))).astype(np.float32))
    with torch.no_grad():
        net1, delta, weight, eta, upmask = m(net, inp, corr, flow, ii, jj)
        disp = torch.as_tensor(rng.uniform(0.3, 2.0, (2, ht, wd, 1)).astype(np.float32))
        up = ref_net.cvx_upsample(disp, upmask[0])
    np.savez_compressed(
        os.path.join(HERE, "update_python.npz"),
        seed=1234, ii=ii.numpy(), jj=jj.numpy(), net=net[0].numpy(), inp=inp[0].numpy(),
        corr=corr[0].numpy(), flow=flow[0].numpy(),
        net1=net1[0].numpy(), delta=delta[0].numpy(), weight=weight[0].numpy(),
        eta=eta[0].numpy(), upmask=upmask[0].numpy().astype(np.float16),
        disp=disp.numpy(), disp_up=up.numpy())
    print("update_python: net1 %s delta %s eta %s upmask %s" % (
        tuple(net1.shape), tuple(delta.shape), tuple(eta.shape), tuple(upmask.shape)))


def golden_update_autocast():
    """The reference's UpdateModule exactly as FactorGraph.update runs it: under torch.autocast (factor_graph.py:13-16,
    214, 227-228) with fp16 hidden state / context / correlation features.  No GPU here, so the autocast device is the
    CPU (same cast policy for conv2d / cat / sigmoid / tanh: fp16 storage of every layer output, fp32 accumulation
    inside a convolution)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_inputs import update_autocast_inputs, UPDATE_AUTOCAST
    torch.manual_seed(0)
    m = ref_net.UpdateModule()
    fill_deterministic(m, seed=UPDATE_AUTOCAST["weight_seed"])
    m.agg.eta[2] = _SoftplusF32()
    m.eval()
    net, inp, corr, flow, ii, jj = update_autocast_inputs()
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
        net1, delta, weight, eta, upmask = m(net[None], inp[None], corr[None], flow[None], ii, jj)
    assert net1.dtype == torch.float16 and upmask.dtype == torch.float16
    np.savez_compressed(
        os.path.join(HERE, "update_autocast_python.npz"),
        net1=net1[0].numpy(), delta=delta[0].float().numpy(), weight=weight[0].float().numpy(),
        eta=eta[0].float().numpy(), upmask=upmask[0].numpy())
    print("update_autocast_python: net1 %s %s delta %s eta %s upmask %s" % (
        tuple(net1.shape), net1.dtype, tuple(delta.shape), tuple(eta.shape), tuple(upmask.shape)))


def golden_encoder():
    """The reference's BasicEncoder (modules/extractor.py) as DroidNet instantiates it (droid_net.py:149-150), under fp16
    autocast like MotionFilter runs it (motion_filter.py:38-49), on two seeded 64x128 images."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_inputs import encoder_inputs
    from modules.extractor import BasicEncoder
    out = {}
    x = encoder_inputs()
    for tag, dim, norm in (("fnet", 128, "instance"), ("cnet", 256, "none")):
        torch.manual_seed(0)
        m = BasicEncoder(output_dim=dim, norm_fn=norm)
        fill_deterministic(m, seed=4321 if tag == "fnet" else 8765)
        m.eval()
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
            y = m(x[None])
        assert y.dtype == torch.float16
        out[tag] = y[0].numpy()
    np.savez_compressed(os.path.join(HERE, "encoder_python.npz"), **out)
    print("encoder_python:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["ba", "corr", "update", "update_autocast", "encoder"]
    for w in which:
        globals()["golden_" + w]()
