"""GPU: droid_amd.factor_graph.FactorGraph / depth_video.DepthVideo / policies (host mirrors over the HIP kernels) against
vectors written by the REFERENCE's own factor_graph.py + depth_video.py + droid_net.py run on CPU with oracle kernels
(tests/golden/graph_python.npz): one composed update iteration incl. inactive edges and upsampling (a1), update_lowmem
(a2), add_proximity_factors with the device NMS (f1), cvx_upsample (f3).  The policy classes (frontend, backend, motion filter,
pose filler) are compared with the reference's own classes in tests/test_policy_gpu.py.

Tolerances: the golden run evaluates the update operator under fp16 autocast and the geometry / BA in fp64; the HIP path
stores fp16 activations and runs the BA in fp32 (fp64 solve).  After one update iteration the hidden state agrees to a few
fp16 roundings (2^-8), flow targets to 1e-2 px, poses to 1e-3, depths to 1e-2 relative; the second iteration starts from
those differences."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import graph as ograph, se3 as ose3, update as oupd
from golden_inputs import graph_scenario


@pytest.fixture(scope="module")
def db():
    assert torch.cuda.is_available()
    import droid_backends
    return droid_backends


class _SD:
    def state_dict(self):
        return oupd.empty_state_dict()


def _setup(S, buffer_extra=2, **kw):
    from droid_amd.depth_video import DepthVideo
    from droid_amd.update import UpdateModule
    from droid_amd.weights import deterministic_state_dict
    N, ht, wd = S["n_frames"], S["ht"], S["wd"]
    video = DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + buffer_extra, device="cuda:0", **kw)
    d = lambda a: torch.as_tensor(a).cuda()
    video.poses[:N] = d(S["poses"]); video.disps[:N] = d(S["disps"]); video.intrinsics[:N] = d(S["intrinsics"])
    video.fmaps[:N, 0] = d(S["fmaps"]); video.nets[:N] = d(S["nets"]); video.inps[:N] = d(S["inps"])
    video.tstamp[:N] = torch.arange(N, device="cuda").float()
    video.counter.value = N
    upd = UpdateModule().load_state_dict(deterministic_state_dict(_SD(), seed=S["weight_seed"]))
    return video, upd


def _rot_angle(q, qr):
    v = q[:, 3:4] * -qr[:, :3] + qr[:, 3:4] * q[:, :3] + np.cross(q[:, :3], -qr[:, :3])
    return 2 * np.linalg.norm(v, axis=-1)


def _compare(video, fg, G, tag, N, scale=1.0, disps_up=True):
    c = lambda t: t.float().cpu().numpy()
    p, rp = c(video.poses)[:N], G[tag + "_poses"][:N]
    assert np.abs(p[:, :3] - rp[:, :3]).max() <= 2e-3 * scale
    assert _rot_angle(p[:, 3:].astype(np.float64), rp[:, 3:].astype(np.float64)).max() <= 2e-3 * scale
    d, rd = c(video.disps)[:N], G[tag + "_disps"][:N]
    e = np.abs(d - rd) / np.maximum(1.0, np.abs(rd))
    assert np.quantile(e, 0.99) <= 1e-2 * scale and e.max() <= 0.1 * scale
    net, rnet = c(fg.net[0]), G[tag + "_net"].astype(np.float32)
    assert np.abs(net - rnet).max() <= 2.0 ** -8 * scale
    tg, rtg = c(fg.target[0]), G[tag + "_target"]
    assert np.quantile(np.abs(tg - rtg), 0.999) <= 2e-2 * scale and np.abs(tg - rtg).max() <= 0.5 * scale
    assert np.abs(c(fg.weight[0]) - G[tag + "_weight"]).max() <= 2.0 ** -8 * scale
    dm, rdm = c(fg.damping)[:N], G[tag + "_damping"][:N]
    assert np.abs(dm - rdm).max() <= 2.0 ** -8 * np.abs(rdm).max() * scale + 1e-6
    if disps_up:
        du, rdu = c(video.disps_up)[:N], G[tag + "_disps_up"][:N].astype(np.float32)
        e = np.abs(du - rdu) / np.maximum(1.0, np.abs(rdu))
        assert np.quantile(e, 0.99) <= 1e-2 * scale + 2.0 ** -10


def test_update_iterations_match_reference_factor_graph(db, golden_dir):
    """scenario A: FactorGraph.update (factor_graph.py:214-263) twice, the second time with three edges moved to the
    inactive set (rm_factors store=True, use_inactive=True) and t0 = 2; upsample=True"""
    from droid_amd.factor_graph import FactorGraph
    G = np.load(os.path.join(golden_dir, "graph_python.npz"))
    S = graph_scenario()
    N = S["n_frames"]
    video, upd = _setup(S)
    fg = FactorGraph(video, upd, corr_impl="volume", max_factors=-1, upsample=True)
    fg.add_neighborhood_factors(0, N, r=2)
    assert np.array_equal(fg.ii.cpu().numpy(), G["A_ii"]) and np.array_equal(fg.jj.cpu().numpy(), G["A_jj"])
    assert np.abs(fg.target[0].cpu().numpy() - G["A_target0"]).max() < 2e-4
    fg.update(t0=1, t1=None, itrs=2, use_inactive=True)
    torch.cuda.synchronize()
    _compare(video, fg, G, "A1", N)
    mask = torch.zeros_like(fg.ii, dtype=torch.bool); mask[:3] = True
    fg.rm_factors(mask, store=True)
    assert np.array_equal(fg.ii.cpu().numpy(), G["A2_ii"]) and np.array_equal(fg.ii_inac.cpu().numpy(), G["A2_ii_inac"])
    fg.update(t0=2, t1=None, itrs=2, use_inactive=True)
    torch.cuda.synchronize()
    _compare(video, fg, G, "A2", N, scale=2.0)
    assert np.array_equal(fg.age.cpu().numpy(), G["A2_age"])


def test_proximity_factors_and_update_lowmem_match_reference(db, golden_dir):
    """scenario B: add_proximity_factors (device NMS) picks the reference's edges in the reference's order, then
    update_lowmem(steps=2) (alt correlation in source-frame chunks + one global BA per step, factor_graph.py:266-330);
    the result does not depend on the chunk size"""
    from droid_amd.factor_graph import FactorGraph
    G = np.load(os.path.join(golden_dir, "graph_python.npz"))
    S = graph_scenario()
    N = S["n_frames"]
    outs = []
    for chunk, corr in ((8, "alt"), (2, "alt"), (8, "pyramid")):       # "pyramid": built once per call, full-batch steps
        video, upd = _setup(S)
        fg = FactorGraph(video, upd, corr_impl="alt", max_factors=16 * N, upsample=False, chunk_frames=chunk)
        dist = video.distance(beta=S["prox_beta"]).reshape(-1).cpu().numpy()
        assert np.abs(dist - G["B_dist"]).max() <= 1e-4 * max(1.0, np.abs(G["B_dist"]).max())
        fg.add_proximity_factors(rad=S["prox_rad"], nms=S["prox_nms"], thresh=S["prox_thresh"], beta=S["prox_beta"])
        assert np.array_equal(fg.ii.cpu().numpy(), G["B_ii"]) and np.array_equal(fg.jj.cpu().numpy(), G["B_jj"])
        fg.update_lowmem(steps=2, corr=corr)
        torch.cuda.synchronize()
        _compare(video, fg, G, "B2", N, scale=2.0, disps_up=False)
        outs.append((video.poses.clone(), video.disps.clone()))
    assert (outs[0][0] - outs[1][0]).abs().max() < 1e-5 and (outs[0][1] - outs[1][1]).abs().max() < 1e-4


def test_incoherent_flow_switches_the_graph_to_reference_layout_volumes(db):
    """FactorGraph.update measures the flow's window spread once per edge list (CorrBlock.window_spread): a reprojection flow keeps
    the pyramid; a flow without spatial coherence (random per-pixel depths under a large baseline: 100-pixel jumps between
    neighbours) is slower in the pyramid layout than in the reference layout, so the graph rebuilds its volumes as
    CorrBlockRef and stays there.  Results = those of a graph that was told native_corr=False from the start."""
    from droid_amd.corr import CorrBlock, CorrBlockRef
    from droid_amd.factor_graph import FactorGraph
    S = graph_scenario()
    N = S["n_frames"]
    # coherent: the scenario's own state
    video, upd = _setup(S)
    fg = FactorGraph(video, upd, corr_impl="volume", max_factors=-1, upsample=False)
    fg.add_neighborhood_factors(0, N, r=2)
    fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
    assert isinstance(fg.corr, CorrBlock) and fg._native_corr and fg.last_window_spread < 2.0
    # incoherent: independent depths per pixel, one-unit baselines
    S2 = dict(S)
    rng = np.random.default_rng(3)
    S2["disps"] = rng.uniform(0.05, 4.0, S["disps"].shape).astype(np.float32)
    poses = S["poses"].copy(); poses[:, 0] = np.arange(N) * 1.0
    S2["poses"] = poses
    outs = []
    for kw in (dict(), dict(native_corr=False)):
        video, upd = _setup(S2)
        fg = FactorGraph(video, upd, corr_impl="volume", max_factors=-1, upsample=False, **kw)
        fg.add_neighborhood_factors(0, N, r=2)
        fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
        torch.cuda.synchronize()
        assert isinstance(fg.corr, CorrBlockRef) and not fg._native_corr
        if not kw:
            assert fg.last_window_spread > CorrBlock.SPREAD_LIMIT
            fg.update(t0=1, t1=None, itrs=2, use_inactive=False)            # stays in the reference layout, no second check
        else:
            fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
        torch.cuda.synchronize()
        outs.append((video.poses[:N].clone(), video.disps[:N].clone(), fg.net.clone(), fg.target.clone()))
    for a, b in zip(*outs):                                   # the same kernels on the same inputs (atomics in the BA: not bit-pinned)
        assert torch.allclose(a.float(), b.float(), atol=1e-5, rtol=1e-5)
    # pinned: native_corr=True keeps the pyramid whatever the flow
    video, upd = _setup(S2)
    fg = FactorGraph(video, upd, corr_impl="volume", max_factors=-1, upsample=False, native_corr=True)
    fg.add_neighborhood_factors(0, N, r=2)
    fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
    assert isinstance(fg.corr, CorrBlock) and fg._native_corr


@pytest.mark.parametrize("case", [dict(t=40, t0=0, t1=0, rad=2, nms=2, thresh=16.0, max_factors=-1, stereo=False),
                                  dict(t=64, t0=10, t1=4, rad=1, nms=1, thresh=12.0, max_factors=300, stereo=True),
                                  dict(t=96, t0=91, t1=71, rad=2, nms=2, thresh=16.0, max_factors=48, stereo=False),
                                  dict(t=30, t0=0, t1=0, rad=3, nms=0, thresh=1e3, max_factors=-1, stereo=False)])
def test_proximity_nms_kernel_vs_oracle(db, case):
    """the greedy NMS walk on random distance matrices incl. existing edges, the max_factors stop and stereo self edges"""
    rng = np.random.default_rng(case["t"])
    t, t0, t1 = case["t"], case["t0"], case["t1"]
    n = (t - t0) * (t - t1)
    d = rng.uniform(0.5, 40.0, n).astype(np.float32)
    d[rng.uniform(size=n) < 0.05] = 1000.0
    ne = 25
    ei = rng.integers(0, t, ne); ej = rng.integers(0, t, ne)
    ref = ograph.proximity_edges(d, t0, t1, t, case["rad"], case["nms"], case["thresh"], case["max_factors"],
                                 list(zip(ei.tolist(), ej.tolist())), case["stereo"])
    n_fixed = sum((1 if case["stereo"] else 0) + 2 * len(range(max(i - case["rad"] - 1, 0), i)) for i in range(t0, t))
    dt = torch.as_tensor(d).cuda()
    out, cnt = db.proximity_nms(dt, torch.as_tensor(ei).cuda(), torch.as_tensor(ej).cuda(), t0, t1, t, case["rad"], case["nms"],
                                case["thresh"], case["max_factors"], n_fixed, case["stereo"], n)
    k = int(cnt.item())
    got = [tuple(r) for r in out[:2 * k].cpu().numpy().tolist()]
    assert got == ref[n_fixed:]


def test_cvx_upsample_and_motion_features(db):
    rng = np.random.default_rng(3)
    K, h, w = 3, 8, 16
    disp = torch.as_tensor(rng.uniform(0.2, 2.0, (K, h, w)).astype(np.float32)).cuda()
    mask = torch.as_tensor((2 * rng.standard_normal((K, 576, h, w))).astype(np.float16)).cuda()
    got = db.cvx_upsample(disp, mask.permute(0, 2, 3, 1).contiguous())
    m16 = torch.softmax(mask.float().view(K, 9, 64, h, w), dim=1).half().float()       # weights rounded to fp16 like the reference's
    ref = oupd.cvx_upsample(disp.cpu()[..., None], torch.log(m16.clamp_min(1e-30)).view(K, 576, h, w).cpu())[..., 0]
    ref32 = oupd.cvx_upsample(disp.cpu()[..., None], mask.float().cpu())[..., 0]
    assert (got.cpu() - ref32).abs().max() <= 2.0 ** -10 * ref32.abs().max()
    E = 4
    c1 = torch.as_tensor(rng.uniform(-80, 150, (E, h, w, 2)).astype(np.float32)).cuda()
    tg = torch.as_tensor(rng.uniform(-80, 150, (E, h, w, 2)).astype(np.float32)).cuda()
    flow = db.motion_features(c1, tg)
    ref = ograph.motion_features(c1.cpu().numpy().astype(np.float64), tg.cpu().numpy().astype(np.float64))
    assert flow.shape == (E, h, w, 8) and torch.count_nonzero(flow[..., 4:]) == 0
    assert np.abs(flow[..., :4].float().cpu().numpy() - ref).max() <= 2.0 ** -10 * 64
    dw = torch.as_tensor(rng.standard_normal((E, h, w, 4)).astype(np.float32)).cuda()
    t, wt, tb, wb = db.ba_inputs(c1, dw)
    assert torch.equal(t, c1 + dw[..., :2]) and torch.equal(wt, dw[..., 2:])
    assert torch.equal(tb, t.permute(0, 3, 1, 2)) and torch.equal(wb, wt.permute(0, 3, 1, 2))


def test_se3_log_and_motion_model(db):
    """lietorch.SE3.log (HIP) vs the oracle, exp(log(T)) = T, and the frontend's damped-velocity prediction"""
    from lietorch import SE3
    rng = np.random.default_rng(9)
    xi = rng.normal(0, 0.5, (200, 6)); xi[0] = 0; xi[1, 3:] = 1e-7
    t, q = ose3.se3_exp(xi)
    T = SE3(torch.as_tensor(ose3.pose_join(t, q).astype(np.float32)).cuda())
    lg = T.log().cpu().numpy()
    assert np.abs(lg - ose3.se3_log(t, q)).max() < 2e-5
    back = SE3.exp(T.log()).data.cpu().numpy()
    sign = np.sign((back[:, 3:] * ose3.pose_join(t, q)[:, 3:]).sum(-1, keepdims=True))
    assert np.abs(back[:, :3] - t).max() < 2e-5 and np.abs(back[:, 3:] * sign - q).max() < 2e-5
    M = T.matrix().cpu().numpy()
    X = rng.normal(0, 1, (200, 3))
    Y = ose3.se3_act(t, q, np.concatenate([X, np.ones((200, 1))], -1))[:, :3]
    assert np.abs((M[:, :3, :3] @ X[..., None])[..., 0] + M[:, :3, 3] - Y).max() < 1e-5


def test_reconstruction_dump_and_point_cloud(db, tmp_path):
    """demo.py:60-76 dump format (keys, shapes, dtypes, round trip) and the viewers' point-cloud extraction
    (view_reconstruction.py:15-38): points of a consistent synthetic scene survive the multi-view filter"""
    from droid_amd.reconstruction import save_reconstruction, load_reconstruction, point_cloud
    from droid_amd import synthetic as syn
    S = graph_scenario()
    N = S["n_frames"]
    video, upd = _setup(S)
    g = syn.small_graph(n_frames=N, seed=13, ht=S["ht"], wd=S["wd"])
    video.poses[:N] = torch.as_tensor(g["poses_gt"]).cuda(); video.disps[:N] = torch.as_tensor(g["disps_gt"]).cuda()
    video.intrinsics[:N] = torch.as_tensor(g["intrinsics"]).cuda()
    video.disps_up[:N] = torch.nn.functional.interpolate(video.disps[:N][None], scale_factor=8, mode="nearest")[0]
    video.images[:N] = torch.randint(0, 255, (N, 3, 8 * S["ht"], 8 * S["wd"]), dtype=torch.uint8, device="cuda")
    path = str(tmp_path / "rec.pth")
    save_reconstruction(video, path)
    blob = load_reconstruction(path)
    assert set(blob) == {"tstamps", "images", "disps", "poses", "intrinsics"}
    assert blob["disps"].shape == (N, 8 * S["ht"], 8 * S["wd"]) and blob["images"].dtype == torch.uint8
    assert torch.equal(blob["poses"], video.poses[:N]) and torch.equal(blob["disps"], video.disps_up[:N])
    pts, cols, mask = point_cloud(video.poses[:N], video.disps[:N], video.intrinsics[0], video.images[:N, :, 3::8, 3::8],
                                  filter_thresh=0.05)
    assert mask.shape == (N, S["ht"], S["wd"]) and pts.shape[1] == 3 and cols.shape == pts.shape
    assert mask.float().mean() > 0.05 and torch.isfinite(pts).all()      # a consistent scene: part of it passes the multi-view check


@pytest.mark.parametrize("k,cin,cout,hw", [(7, 8, 32, (48, 64)), (3, 32, 64, (24, 40)), (3, 64, 128, (16, 16)), (1, 32, 64, (24, 40)), (1, 64, 128, (6, 10))])
def test_stride2_convolution_is_the_stride1_result_at_the_even_positions(db, k, cin, cout, hw):
    """droid_backends.conv2d_s2_nhwc (dh_conv2d_s2_nhwc_f16: the encoders' stem, the first convolution and the 1x1 shortcut of their
    down-sampling blocks, extractor.py:140,24,151) against the stride-1 convolution sliced [::2, ::2] -- what rounds 3-5 computed: same
    taps, same k order per output pixel -> EQUAL bit for bit, linear and relu, several images, image borders"""
    from droid_amd.update import pack_conv, EPI_LINEAR, EPI_RELU
    torch.manual_seed(k * 100 + cin)
    H, W = hw
    x = torch.randn(3, H, W, cin, device="cuda").half()
    w = (torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5)
    b = torch.randn(cout, device="cuda")
    wp, bp = pack_conv(w, b, None)
    for epi in (EPI_LINEAR, EPI_RELU):
        full = torch.empty(3, H, W, cout, dtype=torch.float16, device="cuda")
        db.conv2d_nhwc([x], wp, None, bp, k, k, cout, epi, full, cout, None, None, None, None, None, None, 0, False, 0, False, False)
        got = db.conv2d_s2_nhwc(x, wp, bp, k, k, cout, epi)
        torch.cuda.synchronize()
        assert got.shape == (3, H // 2, W // 2, cout) and torch.equal(got, full[:, ::2, ::2])
    with pytest.raises(RuntimeError):
        db.conv2d_s2_nhwc(x[:, :H - 1].contiguous(), wp, bp, k, k, cout, EPI_LINEAR)            # odd image height


def test_encoders_match_reference_module_under_autocast(db, golden_dir):
    """droid_amd.encoder.BasicEncoder (MFMA convolutions + instance norm / residual kernels) vs vectors written by the
    reference's own BasicEncoder under fp16 autocast (tests/golden/encoder_python.npz); both store fp16 layer outputs"""
    from golden_inputs import encoder_inputs
    from droid_amd.encoder import BasicEncoder, empty_state_dict
    from droid_amd.weights import deterministic_state_dict
    G = np.load(os.path.join(golden_dir, "encoder_python.npz"))
    x = encoder_inputs().cuda()
    for tag, dim, norm, seed in (("fnet", 128, "instance", 4321), ("cnet", 256, "none", 8765)):
        class _S:
            def state_dict(self):
                return empty_state_dict(dim)
        enc = BasicEncoder(dim, norm).load_state_dict(deterministic_state_dict(_S(), seed=seed))
        y = enc(x[None])[0].float().cpu().numpy()
        ref = G[tag].astype(np.float32)
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() <= 2.0 ** -7 * np.abs(ref).max(), tag       # ~20 fp16-stored layers deep
