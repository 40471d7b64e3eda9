"""GPU: the policy mirrors of droid_amd.policies (MotionFilter, PoseTrajectoryFiller, DroidFrontend, DroidBackend) over the HIP
kernels against vectors written by the REFERENCE's own motion_filter.py, trajectory_filler.py, droid_frontend.py and
droid_backend.py run unmodified on CPU (tests/golden/make_policy_golden.py -> tests/golden/policy_python.npz).

Discrete outcomes -- keyframe decisions, the graph's edge lists / ages / inactive sets after every frontend call, the
keyframe-removal branch -- must be EQUAL.  Continuous ones carry the differences of the arithmetic (golden: encoders and
update operator under CPU fp16 autocast, geometry and BA in fp64; HIP: fp16 activations, fp32 BA with an fp64 solve);
make_policy_golden.py --probe shows how far they move when the stored features change by one fp16 ulp (poses 6e-5, depths
2e-3, filled poses 1e-5): the tolerances below leave an order of magnitude above that.
"""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import golden_inputs as gi


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "policy_python.npz"))


@pytest.fixture(scope="module")
def nets_and_update():
    assert torch.cuda.is_available()
    from droid_amd.encoder import FeatureNets, empty_state_dict as enc_sd
    from droid_amd.update import UpdateModule, empty_state_dict as upd_sd
    from droid_amd.weights import deterministic_state_dict

    class _S:
        def __init__(self, sd):
            self.sd = sd

        def state_dict(self):
            return self.sd
    sd = {}
    for pre, dim in (("fnet", 128), ("cnet", 256)):
        sd.update({pre + "." + k: v for k, v in deterministic_state_dict(_S(enc_sd(dim)), seed=gi.POLICY_SEEDS[pre]).items()})
    nets = FeatureNets().load_state_dict(sd)
    upd = UpdateModule().load_state_dict(deterministic_state_dict(_S(upd_sd()), seed=gi.POLICY_SEEDS["update"], scale=2.0))
    return nets, upd


def _scenario_video(S, buffer_extra=24):
    from droid_amd.depth_video import DepthVideo
    N, ht, wd = S["n_frames"], S["ht"], S["wd"]
    video = DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + buffer_extra, device="cuda:0")
    d = lambda a: torch.as_tensor(a).cuda()
    video.poses[:N] = d(S["poses"]); video.disps[:N] = d(S["disps"]); video.intrinsics[:N] = d(S["intrinsics"])
    video.fmaps[:N, 0] = d(S["fmaps"]); video.nets[:N] = d(S["nets"]); video.inps[:N] = d(S["inps"])
    video.tstamp[:N] = torch.arange(N, device="cuda").float()
    video.counter.value = N
    return video


def _rot_angle(q, qr):
    v = q[:, 3:4] * -qr[:, :3] + qr[:, 3:4] * q[:, :3] + np.cross(q[:, :3], -qr[:, :3])
    return 2 * np.linalg.norm(v, axis=-1)


def _poses_close(p, rp, tol):
    assert np.abs(p[:, :3] - rp[:, :3]).max() <= tol
    assert _rot_angle(p[:, 3:].astype(np.float64), rp[:, 3:].astype(np.float64)).max() <= tol


def test_motion_filter_decisions_and_flow_magnitude_match_reference(G, nets_and_update):
    """MotionFilter.track (motion_filter.py:52-91) on a 10-frame pan: the mean flow magnitude of the single update iteration
    per frame (delta.norm(dim=-1).mean()), the keyframe decision it drives and the features stored per accepted keyframe"""
    from droid_amd.depth_video import DepthVideo
    from droid_amd.policies import MotionFilter
    nets, upd = nets_and_update
    ht, wd = gi.POLICY_IMAGE
    video = DepthVideo(image_size=[ht, wd], buffer=16, device="cuda:0")
    mf = MotionFilter(nets, upd, video, thresh=gi.MOTION_FILTER_THRESH)
    intr = torch.tensor(gi.MOTION_FILTER_INTRINSICS)
    deltas, counters = [], []
    for k, s in enumerate(gi.MOTION_FILTER_SHIFTS):
        mf.track(float(k), gi.policy_image(7, s), intrinsics=intr)
        counters.append(video.counter.value)
        if k > 0:
            deltas.append(mf.last_delta)
    assert counters == G["M_counter"].tolist()                                  # the same frames became keyframes
    # the golden's value is an fp16 mean (CPU autocast): spacing 2^-10 below 1
    assert np.abs(np.array(deltas) - G["M_delta"]).max() <= 2.5e-3
    assert min(abs(d - gi.MOTION_FILTER_THRESH) for d in G["M_delta"]) > 2 * 2.5e-3     # decisions are not marginal
    n = video.counter.value
    c = lambda t: t.float().cpu().numpy()
    assert np.array_equal(c(video.tstamp[:n]), G["M_tstamp"]) and np.abs(c(video.intrinsics[:n]) - G["M_intrinsics"]).max() < 1e-6
    for name in ("fmaps", "nets", "inps"):
        got, ref = c(getattr(video, name)[:n]), G["M_" + name].astype(np.float32)
        assert np.abs(got - ref).max() <= 2.0 ** -7 * np.abs(ref).max(), name       # ~20 fp16-stored encoder layers deep


def test_pose_trajectory_filler_matches_reference(G, nets_and_update):
    """PoseTrajectoryFiller.__call__ (trajectory_filler.py:42-111) on 18 non-keyframes (one batch of 16 + 2): SE(3)
    interpolation between the bracketing keyframes, feature encoder, two edges per frame, six motion-only update
    iterations; keyframes stay untouched"""
    from droid_amd.policies import PoseTrajectoryFiller
    nets, upd = nets_and_update
    S = gi.graph_scenario()
    N = S["n_frames"]
    video = _scenario_video(S)
    poses0, disps0 = video.poses.clone(), video.disps.clone()
    out = PoseTrajectoryFiller(nets, upd, video)(gi.filler_stream())
    torch.cuda.synchronize()
    assert video.counter.value == N and torch.equal(video.poses[:N], poses0[:N]) and torch.equal(video.disps[:N], disps0[:N])
    got = out.data.cpu().numpy()
    assert got.shape == G["T_poses"].shape == (18, 7)
    _poses_close(got, G["T_poses"], 1e-3)


def test_backend_matches_reference(G, nets_and_update):
    """DroidBackend.__call__ (droid_backend.py:24-42): normalize -> proximity edges -> two global-BA steps -> clear_edges"""
    from types import SimpleNamespace
    from droid_amd.policies import DroidBackend
    nets, upd = nets_and_update
    S = gi.graph_scenario()
    N = S["n_frames"]
    for corr in ("alt", "pyramid"):
        video = _scenario_video(S)
        be = DroidBackend(upd, video, SimpleNamespace(**gi.BACKEND_ARGS), chunk_frames=8)
        be.lowmem_corr = corr
        g = be(steps=2)
        torch.cuda.synchronize()
        assert len(g.ii) == 0
        _poses_close(video.poses[:N].cpu().numpy(), G["K_poses"], 2e-3)
        d, rd = video.disps[:N].cpu().numpy(), G["K_disps"]
        e = np.abs(d - rd) / np.maximum(1.0, np.abs(rd))
        assert np.quantile(e, 0.99) <= 1e-2 and e.max() <= 0.1


def test_frontend_matches_reference_call_by_call(G, nets_and_update):
    """DroidFrontend (droid_frontend.py:65-164) on a 14-frame synthetic sequence: initialisation (8 + 8 update iterations
    around add_proximity_factors) and six keyframe updates, one of which removes a keyframe (rm_keyframe) while the others
    run the two extra iterations.  After EVERY call: t1 / counter, the edge lists, ages and inactive sets equal the
    reference's; poses / depths within the arithmetic's tolerance."""
    from types import SimpleNamespace
    from droid_amd.policies import DroidFrontend
    nets, upd = nets_and_update
    S = gi.graph_scenario(n_frames=14)
    video = _scenario_video(S)
    fe = DroidFrontend(upd, video, SimpleNamespace(**gi.FRONTEND_ARGS))
    dist_log = []
    vd = video.distance

    def recording_distance(ii=None, jj=None, beta=0.3, bidirectional=True):
        d = vd(ii, jj, beta=beta, bidirectional=bidirectional)
        if isinstance(ii, list) and len(ii) == 1:
            dist_log.append(float(d.item()))
        return d
    video.distance = recording_distance

    def put(k, src):
        d = lambda a: torch.as_tensor(a).cuda()
        video.tstamp[k] = float(src); video.intrinsics[k] = d(S["intrinsics"][src])
        video.fmaps[k, 0] = d(S["fmaps"][src]); video.nets[k] = d(S["nets"][src]); video.inps[k] = d(S["inps"][src])

    seen = []

    def snapshot(tag):
        torch.cuda.synchronize()
        g = fe.graph
        t = video.counter.value
        assert [fe.t1, t] == G["F_%s_state" % tag].tolist(), tag
        for name in ("ii", "jj", "age", "ii_inac", "jj_inac", "ii_bad", "jj_bad"):
            assert np.array_equal(getattr(g, name).cpu().numpy(), G["F_%s_%s" % (tag, name)]), (tag, name)
        assert np.array_equal(video.tstamp[:t].cpu().numpy(), G["F_%s_tstamp" % tag]), tag
        _poses_close(video.poses[:t + 1].cpu().numpy(), G["F_%s_poses" % tag], 2e-3)
        d, rd = video.disps[:t + 1].cpu().numpy(), G["F_%s_disps" % tag]
        e = np.abs(d - rd) / np.maximum(1.0, np.abs(rd))
        assert np.quantile(e, 0.99) <= 2e-2 and e.max() <= 0.2, (tag, np.quantile(e, 0.99), e.max())
        seen.append(tag)
    gi.drive_frontend(video, fe, S, put, snapshot)
    assert seen == ["init"] + ["f%d" % k for k in range(9, 15)]
    ref_d = G["F_keyframe_distance"]
    assert len(dist_log) == len(ref_d) and np.abs(np.array(dist_log) - ref_d).max() <= 2e-2 * max(1.0, np.abs(ref_d).max())
    thr = 2 * gi.FRONTEND_ARGS["keyframe_thresh"]
    assert (ref_d < thr).any() and (ref_d >= thr).any()                         # both branches of the keyframe-removal test ran
