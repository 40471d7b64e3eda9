"""Seeded inputs shared by tests/golden/make_golden.py (which runs the REFERENCE on them) and the tests (which run the
oracle / the HIP path on the same values): keeps the committed fixtures to the reference's OUTPUTS."""
import numpy as np
import torch

UPDATE_AUTOCAST = dict(E=3, ht=8, wd=64, seed=41, ii=[0, 0, 2], jj=[1, 2, 0], weight_seed=1234)


def update_autocast_inputs():
    """fp16-representable inputs of one UpdateModule.forward as FactorGraph.update feeds it under autocast
    (factor_graph.py:214-231): net, inp, corr in fp16, motion features fp32 clamped to +-64.  W = 64 and H % 4 == 0, so
    the HIP path takes its production convolution kernels."""
    c = UPDATE_AUTOCAST
    rng = np.random.default_rng(c["seed"])
    E, ht, wd = c["E"], c["ht"], c["wd"]
    net = np.tanh(rng.standard_normal((E, 128, ht, wd))).astype(np.float16)
    inp = np.maximum(rng.standard_normal((E, 128, ht, wd)), 0).astype(np.float16)
    corr = (2.0 * rng.standard_normal((E, 196, ht, wd))).astype(np.float16)
    flow = np.clip(4.0 * rng.standard_normal((E, 4, ht, wd)), -64, 64).astype(np.float16).astype(np.float32)
    t = lambda a: torch.as_tensor(a)
    return t(net), t(inp), t(corr), t(flow), torch.as_tensor(c["ii"]), torch.as_tensor(c["jj"])


def graph_scenario(n_frames=6, ht=16, wd=64):
    """Six keyframes at 16 x 64 (1/8 resolution; W = 64 so the HIP path takes its production kernels): state of a
    DepthVideo (poses, depths, features) for the factor-graph golden run (tests/golden/make_graph_golden.py).
    (ht, wd) = (30, 40) is TUM's 240 x 320: outside the pyramid layout and the production convolution tiling, i.e. the
    generic paths (reference-layout volumes, generic convolution loop)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "droid-slam_amd"))
    from droid_amd import synthetic as syn
    N = n_frames
    rng = np.random.default_rng(77)
    cfg = syn.GraphConfig("scn", N, 2 * (N - 1) + 2 * (N - 2), radius=2)
    g = syn.make_graph(cfg, seed=5, ht=ht, wd=wd)
    intr = np.array([24.0, 24.0, wd / 2.0, ht / 2.0], dtype=np.float32)
    disps = (0.7 + 0.6 * g["disps_gt"] / g["disps_gt"].max()).astype(np.float32)
    return dict(ht=ht, wd=wd, n_frames=N, weight_seed=1234,
                poses=g["poses"].astype(np.float32), disps=disps, intrinsics=np.tile(intr, (N, 1)),
                fmaps=rng.standard_normal((N, 128, ht, wd)).astype(np.float16),
                nets=np.tanh(rng.standard_normal((N, 128, ht, wd))).astype(np.float16),
                inps=np.maximum(rng.standard_normal((N, 128, ht, wd)), 0).astype(np.float16),
                prox_rad=1, prox_nms=1, prox_thresh=1e3, prox_beta=0.25)


def encoder_inputs():
    """two normalised 64 x 128 images [2,3,64,128] float32 (seeded)"""
    rng = np.random.default_rng(123)
    img = rng.integers(0, 256, (2, 3, 64, 128)).astype(np.float32) / 255.0
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[:, None, None]
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[:, None, None]
    return torch.as_tensor((img - mean) / std)


# edges of the C2 graph (512 edges) whose full hidden state / target / weight is kept in graph_c2_python.npz
C2_SAMPLE_EDGES = [0, 1, 2, 3, 254, 255, 256, 257, 508, 509, 510, 511]

# edges of the C3 graph (4096 edges) whose target / weight / subsampled hidden state is kept in graph_c3_python.npz: every
# 65th edge (hits temporal and loop-closure edges, all strips of the lookup, all frame groups of the update operator)
C3_SAMPLE_EDGES = list(range(0, 4096, 65))[:63] + [4095]
C3_SAMPLE_FRAMES = list(range(0, 512, 8))
# C5 (1024 keyframes / 8192 edges, stereo): 64 edges spread over the list (the first 1024 are the stereo self-edges) + 64 frames
C5_SAMPLE_EDGES = list(range(0, 8192, 130))[:63] + [8191]
C5_SAMPLE_FRAMES = list(range(0, 1024, 16))


def stereo_scenario(n_frames=6):
    """Six stereo keyframes at 16 x 64: feature maps of both cameras, stereo self-edges (i, i) followed by the temporal
    edges |i - j| <= 2 (tests/golden/make_graph_scale_golden.py, scenario S)."""
    S = graph_scenario(n_frames)
    N, ht, wd = S["n_frames"], S["ht"], S["wd"]
    rng = np.random.default_rng(78)
    right = rng.standard_normal((N, 128, ht, wd)).astype(np.float16)
    S["fmaps"] = np.stack([S["fmaps"], right], 1)                       # [N, 2, 128, h, w]
    es = [(i, i) for i in range(N)] + [(i, j) for i in range(N) for j in range(N) if i != j and abs(i - j) <= 2]
    S["ii"] = np.array([e[0] for e in es], dtype=np.int64); S["jj"] = np.array([e[1] for e in es], dtype=np.int64)
    return S


# ---------------------------------------------------------------------------------------------- policies (f1 / f2 / f3)
POLICY_SEEDS = dict(fnet=11, cnet=12, update=1234)
POLICY_IMAGE = (128, 512)                        # 16 x 64 at 1/8 resolution: the production kernels of the HIP path


def policy_image(seed, shift=0):
    """smooth random BGR image [1,3,128,512] uint8 (numpy only: identical on every host), shifted right by `shift` pixels"""
    ht, wd = POLICY_IMAGE
    rng = np.random.default_rng(seed)
    img = np.kron(rng.uniform(0, 255, (3, ht // 8, wd // 8 + 16)), np.ones((8, 8)))
    for ax in (1, 2):                                    # 9-tap box filter along both axes (cumulative sums)
        c = np.cumsum(np.pad(img, [(0, 0)] + [(5, 4) if a == ax else (0, 0) for a in (1, 2)], mode="edge"), axis=ax)
        img = (np.take(c, np.arange(9, c.shape[ax]), axis=ax) - np.take(c, np.arange(0, c.shape[ax] - 9), axis=ax)) / 9.0
    img = img[:, :, 128 - shift:128 - shift + wd]
    return torch.as_tensor(np.clip(np.rint(img), 0, 255).astype(np.uint8))[None]


MOTION_FILTER_SHIFTS = [0, 0, 2, 10, 11, 24, 24, 40, 41, 72]      # camera pans: pixels of image shift per incoming frame
MOTION_FILTER_INTRINSICS = [200.0, 200.0, 256.0, 64.0]


def filler_stream(n=18):
    """image stream of the pose filler: (tstamp, image [1,3,H,W] uint8, intrinsics [4]) for n non-keyframes between / at /
    beyond the keyframe time stamps 0..5 (18 frames: one full batch of 16 + a remainder, trajectory_filler.py:96-107)"""
    ts = np.round(np.linspace(0.0, 5.4, n) + 0.013 * np.arange(n), 3)
    ts[4] = 2.0                                           # exactly at a keyframe
    return [(float(t), policy_image(100 + k), torch.tensor(MOTION_FILTER_INTRINSICS)) for k, t in enumerate(ts)]


FRONTEND_ARGS = dict(upsample=True, warmup=8, beta=0.3, frontend_nms=1, keyframe_thresh=0.375, frontend_window=20, frontend_thresh=16.0,
                     frontend_radius=2)


def drive_frontend(video, fe, pool, put, snapshot):
    """the frontend's life on a synthetic sequence: `warmup` keyframes present -> initialisation; then one keyframe at a time
    is appended behind the last one (what MotionFilter.track does: features / time stamp of the next frame of `pool`, pose
    and depth guesses left as the frontend set them) and the frontend is called.  Runs unchanged against the reference's
    DroidFrontend (golden) and droid_amd.policies.DroidFrontend (test)."""
    nxt = FRONTEND_ARGS["warmup"]
    video.counter.value = nxt
    fe()
    snapshot("init")
    while nxt < pool["n_frames"]:
        k = video.counter.value
        put(k, nxt)
        video.counter.value = k + 1
        nxt += 1
        fe()
        snapshot("f%d" % nxt)
MOTION_FILTER_THRESH = 0.905
BACKEND_ARGS = dict(upsample=False, beta=0.25, backend_thresh=1e3, backend_radius=1, backend_nms=1)
