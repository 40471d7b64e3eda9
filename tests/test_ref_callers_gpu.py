"""GPU: the REFERENCE's own callers of the path -- droid_slam/factor_graph.py, depth_video.py, droid_net.py, modules/corr.py,
geom/projective_ops.py, byte for byte (oracle/_ref/ref_py.zip, staged by oracle/build_ref.py; python imports straight from the
archive) -- executed UNCHANGED on this repository's `droid_backends` (HIP kernels), `lietorch` and `torch_scatter`:
north_star's "exposed through the existing droid_backends PyTorch extension API so factor_graph.py / depth_video.py call it
unchanged".  The update operator is the reference's torch module (nn.Conv2d under fp16 autocast on the ROCm device).

Checked against (i) the golden of the same scenario written by the same reference files on CPU with oracle kernels
(tests/golden/graph_python.npz) and (ii) this repository's mirrors (droid_amd.FactorGraph + the HIP update operator).
"""
import os
import sys
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "droid-slam_amd")
ZIP = os.path.join(ROOT, "oracle", "_ref", "ref_py.zip")

from golden_inputs import graph_scenario


@pytest.fixture(scope="module")
def refpy():
    assert torch.cuda.is_available()
    if not os.path.exists(ZIP):
        pytest.skip("oracle/_ref/ref_py.zip not staged (oracle/build_ref.py needs /root/reference)")
    import droid_backends, lietorch, torch_scatter                      # this repository's (conftest puts droid-slam_amd/ first)
    for m in (droid_backends, lietorch, torch_scatter):
        assert m.__file__.startswith(PKG), m.__file__
    p = os.path.join(ZIP, "droid_slam")
    if p not in sys.path:
        sys.path.append(p)                                              # behind everything of this repository
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import factor_graph as ref_fg, depth_video as ref_dv, droid_net as ref_net
    for m in (ref_fg, ref_dv, ref_net):
        assert ZIP in m.__file__, m.__file__
    import modules.corr as ref_corr
    assert ref_corr.droid_backends is droid_backends                    # the reference's CorrBlock calls the HIP kernels
    return ref_fg, ref_dv, ref_net


def _ref_video(ref_dv, S, **kw):
    N, ht, wd = S["n_frames"], S["ht"], S["wd"]
    video = ref_dv.DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, device="cuda:0", **kw)
    d = lambda a: torch.as_tensor(a).cuda()
    video.poses[:N] = d(S["poses"]); video.disps[:N] = d(S["disps"]); video.intrinsics[:N] = d(S["intrinsics"])
    video.fmaps[:N, 0] = d(S["fmaps"]); video.nets[:N] = d(S["nets"]); video.inps[:N] = d(S["inps"])
    video.counter.value = N
    return video


def _rot_angle(q, qr):
    v = q[:, 3:4] * -qr[:, :3] + qr[:, 3:4] * q[:, :3] + np.cross(q[:, :3], -qr[:, :3])
    return 2 * np.linalg.norm(v, axis=-1)


def _compare(video, fg, G, tag, N, scale=1.0, disps_up=True):
    """the tolerances of tests/test_graph_gpu.py::_compare (HIP mirrors vs the same golden)"""
    c = lambda t: t.float().cpu().numpy()
    p, rp = c(video.poses)[:N], G[tag + "_poses"][:N]
    assert np.abs(p[:, :3] - rp[:, :3]).max() <= 2e-3 * scale
    assert _rot_angle(p[:, 3:].astype(np.float64), rp[:, 3:].astype(np.float64)).max() <= 2e-3 * scale
    d, rd = c(video.disps)[:N], G[tag + "_disps"][:N]
    e = np.abs(d - rd) / np.maximum(1.0, np.abs(rd))
    assert np.quantile(e, 0.99) <= 1e-2 * scale and e.max() <= 0.1 * scale
    assert np.abs(c(fg.net[0]) - G[tag + "_net"].astype(np.float32)).max() <= 2.0 ** -8 * scale
    tg, rtg = c(fg.target[0]), G[tag + "_target"]
    assert np.quantile(np.abs(tg - rtg), 0.999) <= 2e-2 * scale and np.abs(tg - rtg).max() <= 0.5 * scale
    assert np.abs(c(fg.weight[0]) - G[tag + "_weight"]).max() <= 2.0 ** -8 * scale
    dm, rdm = c(fg.damping)[:N], G[tag + "_damping"][:N]
    assert np.abs(dm - rdm).max() <= 2.0 ** -8 * np.abs(rdm).max() * scale + 1e-6
    if disps_up:
        du, rdu = c(video.disps_up)[:N], G[tag + "_disps_up"][:N].astype(np.float32)
        e = np.abs(du - rdu) / np.maximum(1.0, np.abs(rdu))
        assert np.quantile(e, 0.99) <= 1e-2 * scale + 2.0 ** -10


def _update_module(ref_net, seed):
    from droid_amd.weights import fill_deterministic
    m = ref_net.UpdateModule()
    fill_deterministic(m, seed=seed)
    return m.cuda().eval()


def test_reference_factor_graph_update_runs_unchanged_on_the_hip_backend(refpy, golden_dir):
    """scenario A of tests/golden/make_graph_golden.py: reference FactorGraph.add_neighborhood_factors + update twice (the
    second with inactive edges), through reference DepthVideo.reproject (this repo's lietorch), reference CorrBlock
    (droid_backends.corr_index_forward = HIP), reference DepthVideo.ba (droid_backends.ba = HIP)"""
    ref_fg, ref_dv, ref_net = refpy
    G = np.load(os.path.join(golden_dir, "graph_python.npz"))
    S = graph_scenario()
    N = S["n_frames"]
    video = _ref_video(ref_dv, S)
    with torch.no_grad():
        fg = ref_fg.FactorGraph(video, _update_module(ref_net, S["weight_seed"]), device="cuda:0", corr_impl="volume", max_factors=-1, upsample=True)
        fg.add_neighborhood_factors(0, N, r=2)
        assert np.array_equal(fg.ii.cpu().numpy(), G["A_ii"]) and np.array_equal(fg.jj.cpu().numpy(), G["A_jj"])
        assert np.abs(fg.target[0].cpu().numpy() - G["A_target0"]).max() < 2e-4
        fg.update(t0=1, t1=None, itrs=2, use_inactive=True)
        torch.cuda.synchronize()
        _compare(video, fg, G, "A1", N)
        state_ref = (video.poses[:N].clone(), video.disps[:N].clone(), fg.net.clone(), fg.target.clone())
        mask = torch.zeros_like(fg.ii, dtype=torch.bool); mask[:3] = True
        fg.rm_factors(mask, store=True)
        fg.update(t0=2, t1=None, itrs=2, use_inactive=True)
        torch.cuda.synchronize()
        _compare(video, fg, G, "A2", N, scale=2.0)
    # the same first iteration through this repository's mirrors: reference callers and mirrors agree on the device
    from droid_amd.depth_video import DepthVideo
    from droid_amd.factor_graph import FactorGraph
    from droid_amd.update import UpdateModule, empty_state_dict
    from droid_amd.weights import deterministic_state_dict

    class _SD:
        def state_dict(self):
            return empty_state_dict()
    mv = DepthVideo(image_size=[8 * S["ht"], 8 * S["wd"]], buffer=N + 2, device="cuda:0")
    d = lambda a: torch.as_tensor(a).cuda()
    mv.poses[:N] = d(S["poses"]); mv.disps[:N] = d(S["disps"]); mv.intrinsics[:N] = d(S["intrinsics"])
    mv.fmaps[:N, 0] = d(S["fmaps"]); mv.nets[:N] = d(S["nets"]); mv.inps[:N] = d(S["inps"]); mv.counter.value = N
    mfg = FactorGraph(mv, UpdateModule().load_state_dict(deterministic_state_dict(_SD(), seed=S["weight_seed"])), corr_impl="volume", upsample=True)
    mfg.add_neighborhood_factors(0, N, r=2)
    mfg.update(t0=1, t1=None, itrs=2, use_inactive=True)
    torch.cuda.synchronize()
    assert (mv.poses[:N, :3] - state_ref[0][:, :3]).abs().max().item() <= 2e-3
    e = ((mv.disps[:N] - state_ref[1]).abs() / state_ref[1].abs().clamp(min=1.0)).flatten()
    assert torch.quantile(e, 0.99).item() <= 1e-2
    assert (mfg.net.float() - state_ref[2].float()).abs().max().item() <= 2.0 ** -8
    assert torch.quantile((mfg.target - state_ref[3]).abs().flatten(), 0.999).item() <= 2e-2


def test_reference_global_ba_runs_unchanged_on_the_hip_backend(refpy, golden_dir):
    """scenario B: reference add_proximity_factors (droid_backends.frame_distance = HIP, the reference's own CPU NMS loop)
    and update_lowmem (reference AltCorrBlock -> droid_backends.altcorr_forward = HIP, one global ba per step)"""
    ref_fg, ref_dv, ref_net = refpy
    G = np.load(os.path.join(golden_dir, "graph_python.npz"))
    S = graph_scenario()
    N = S["n_frames"]
    video = _ref_video(ref_dv, S)
    with torch.no_grad():
        fg = ref_fg.FactorGraph(video, _update_module(ref_net, S["weight_seed"]), device="cuda:0", corr_impl="alt", max_factors=16 * N, upsample=False)
        fg.add_proximity_factors(rad=S["prox_rad"], nms=S["prox_nms"], thresh=S["prox_thresh"], beta=S["prox_beta"])
        assert np.array_equal(fg.ii.cpu().numpy(), G["B_ii"]) and np.array_equal(fg.jj.cpu().numpy(), G["B_jj"])
        fg.update_lowmem(steps=2)
        torch.cuda.synchronize()
    _compare(video, fg, G, "B2", N, scale=2.0, disps_up=False)
