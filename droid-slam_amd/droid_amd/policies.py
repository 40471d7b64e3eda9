"""The callers of the update iteration: local-BA frontend, global-BA backend, keyframe selection, non-keyframe pose filler.

Host-side mirrors of the reference's policy classes -- DroidFrontend (droid_slam/droid_frontend.py:13-164), DroidBackend
(droid_slam/droid_backend.py:9-43), MotionFilter (droid_slam/motion_filter.py:20-91) and PoseTrajectoryFiller
(droid_slam/trajectory_filler.py:21-111) -- on top of droid_amd.factor_graph.FactorGraph / droid_amd.depth_video.DepthVideo /
droid_amd.encoder.FeatureNets.  Same hyper-parameters, same order of graph edits and update calls; the work itself
(encoders, distances, NMS, update iterations, BA, SE(3) log/exp of the motion model) runs in the kernels of libdroid_hip.
Pinned to the reference's own classes by tests/golden/policy_python.npz (tests/test_policy_gpu.py).
"""
import torch

from lietorch import SE3
from .factor_graph import FactorGraph


class DroidFrontend:
    def __init__(self, update_op, video, args):
        self.video = video
        self.update_op = update_op
        self.graph = FactorGraph(video, update_op, max_factors=48, upsample=getattr(args, "upsample", False))
        self.t0 = 0
        self.t1 = 0
        self.is_initialized = False
        self.count = 0
        self.max_age = 20
        self.iters1 = 3
        self.iters2 = 2
        self.keyframe_removal_index = 3
        self.warmup = args.warmup
        self.beta = args.beta
        self.frontend_nms = args.frontend_nms
        self.keyframe_thresh = args.keyframe_thresh
        self.frontend_window = args.frontend_window
        self.frontend_thresh = args.frontend_thresh
        self.frontend_radius = args.frontend_radius
        self.depth_window = 3
        self.motion_damping = getattr(args, "motion_damping", 0.0)

    def _init_next_state(self):
        """pose / depth guess for the next keyframe: damped constant-velocity model (droid_frontend.py:50-63)"""
        v, t1 = self.video, self.t1
        v.poses[t1] = v.poses[t1 - 1]
        v.disps[t1] = torch.quantile(v.disps[t1 - 3:t1 - 1], 0.5)
        if self.motion_damping >= 0:
            poses = SE3(v.poses)
            vel = (poses[t1 - 1] * poses[t1 - 2].inv()).log()
            v.poses[t1] = (SE3.exp(self.motion_damping * vel) * poses[t1 - 1]).data

    def _update(self):
        """new keyframe: edit the graph, run the update iterations, maybe drop a redundant keyframe (droid_frontend.py:65-117)"""
        v = self.video
        self.count += 1
        self.t1 += 1
        if self.graph.corr is not None:
            self.graph.rm_factors(self.graph.age > self.max_age, store=True)
        self.graph.add_proximity_factors(self.t1 - 5, max(self.t1 - self.frontend_window, 0), rad=self.frontend_radius,
                                         nms=self.frontend_nms, thresh=self.frontend_thresh, beta=self.beta, remove=True)
        v.disps[self.t1 - 1] = torch.where(v.disps_sens[self.t1 - 1] > 0, v.disps_sens[self.t1 - 1], v.disps[self.t1 - 1])
        for _ in range(self.iters1):
            self.graph.update(None, None, use_inactive=True)
        d = v.distance([self.t1 - 4], [self.t1 - 2], beta=self.beta, bidirectional=True)
        if d.item() < 2 * self.keyframe_thresh:
            self.graph.rm_keyframe(self.t1 - 3)
            with v.get_lock():
                v.counter.value -= 1
                self.t1 -= 1
        else:
            for _ in range(self.iters2):
                self.graph.update(None, None, use_inactive=True)
        v.poses[self.t1] = v.poses[self.t1 - 1]
        v.disps[self.t1] = torch.quantile(v.disps[self.t1 - self.depth_window - 1:self.t1 - 1], 0.7)
        v.dirty[self.graph.ii.min():self.t1] = True

    def _initialize(self):
        """first `warmup` keyframes: neighbourhood edges, 8 + 8 update iterations (droid_frontend.py:119-151)"""
        v = self.video
        self.t0 = 0
        self.t1 = v.counter.value
        self.graph.add_neighborhood_factors(self.t0, self.t1, r=3)
        for _ in range(8):
            self.graph.update(1, use_inactive=True)
        self.graph.add_proximity_factors(0, 0, rad=2, nms=2, thresh=self.frontend_thresh, remove=False)
        for _ in range(8):
            self.graph.update(1, use_inactive=True)
        v.poses[self.t1] = v.poses[self.t1 - 1].clone()
        v.disps[self.t1] = v.disps[self.t1 - 4:self.t1].mean()
        self.is_initialized = True
        self.last_pose = v.poses[self.t1 - 1].clone()
        self.last_disp = v.disps[self.t1 - 1].clone()
        self.last_time = v.tstamp[self.t1 - 1].clone()
        with v.get_lock():
            v.ready.value = 1
            v.dirty[:self.t1] = True
        self.graph.rm_factors(self.graph.ii < self.warmup - 4, store=True)

    def __call__(self):
        if not self.is_initialized and self.video.counter.value == self.warmup:
            self._initialize()
            self._init_next_state()
        elif self.is_initialized and self.t1 < self.video.counter.value:
            self._update()
            self._init_next_state()


class DroidBackend:
    """global bundle adjustment over all keyframes (droid_backend.py:9-43)"""

    def __init__(self, update_op, video, args, chunk_frames=8, graph_cls=FactorGraph, graph_kwargs=None):
        """graph_cls / graph_kwargs: droid_amd.dist_graph.DistFactorGraph (+ group=...) runs the same global BA partitioned by
        edge batches over the ranks of a process group (BASELINE configs[3] / [4]); every rank calls the backend together."""
        self.graph_cls, self.graph_kwargs = graph_cls, dict(graph_kwargs or {})
        self._arena = None                 # the pyramid's storage, kept from one global BA to the next (a cold 105 GB hipMalloc takes seconds)
        self.video = video
        self.update_op = update_op
        self.upsample = getattr(args, "upsample", False)
        self.beta = args.beta
        self.backend_thresh = args.backend_thresh
        self.backend_radius = args.backend_radius
        self.backend_nms = args.backend_nms
        self.chunk_frames = chunk_frames
        self.lowmem_corr = "auto"          # FactorGraph.update_lowmem: pyramid built once per call when it fits, else alt-corr chunks

    @torch.no_grad()
    def __call__(self, steps=12, normalize=True):
        v = self.video
        t = v.counter.value
        if normalize and not v.stereo and not torch.any(v.disps_sens):
            v.normalize()
        graph = self.graph_cls(v, self.update_op, corr_impl="alt", max_factors=16 * t, upsample=self.upsample, chunk_frames=self.chunk_frames,
                               **self.graph_kwargs)
        graph.add_proximity_factors(rad=self.backend_radius, nms=self.backend_nms, thresh=self.backend_thresh, beta=self.beta)
        graph._arena = self._arena
        try:
            graph.update_lowmem(steps=steps, corr=self.lowmem_corr)
        finally:
            self._arena = getattr(graph, "_arena", None)      # None after an out-of-memory fallback: the storage was released
            graph.release_pyramid_arena()                     # the returned graph does not pin ~100 GB
        graph.clear_edges()
        v.dirty[:t] = True
        return graph

    def release(self):
        """give the pyramid storage back to the allocator (e.g. before the frontend needs the memory)"""
        self._arena = None


class MotionFilter:
    """keyframe selection (motion_filter.py:20-91): encode every incoming frame, estimate the flow to the last keyframe
    with ONE update iteration (no BA) on a single self-built correlation volume, append the frame to the video if the mean
    flow magnitude exceeds `thresh`."""

    def __init__(self, nets, update_op, video, thresh=2.5):
        self.nets, self.update, self.video, self.thresh = nets, update_op, video, thresh
        self.count = 0
        self.last_delta = float("nan")

    @torch.no_grad()
    def track(self, tstamp, image, depth=None, intrinsics=None):
        """image [1,3,H,W] uint8 BGR (one camera) or [2,3,H,W] (stereo pair)"""
        from .corr import CorrBlock, CorrBlockRef
        v = self.video
        dev = v.device
        ht, wd = image.shape[-2] // 8, image.shape[-1] // 8
        image = image.to(dev)
        x = image[None]
        from .encoder import normalize_images
        xn = normalize_images(x)
        gmap = self.nets.fnet(xn)[0]                                          # [rig,128,h,w]
        Id = torch.tensor([0, 0, 0, 0, 0, 0, 1.0], device=dev)
        intr8 = None if intrinsics is None else intrinsics.to(dev) / 8.0

        def context():
            net, inp = self.nets.cnet(xn[:, [0]])[0].split([128, 128], dim=1)
            return torch.tanh(net.float()).half(), torch.relu(inp)
        if v.counter.value == 0:
            self.net, self.inp = context()
            self.fmap = gmap
            # (the reference stores net[0,0] / inp[0,0] here, motion_filter.py:74: channel 0 of the context maps, broadcast
            # over all 128 channels of keyframe 0 by the buffer assignment; later keyframes get all channels, :88.  Kept.)
            v.append(tstamp, image[0], Id, 1.0, depth, intr8, gmap, self.net[0, 0], self.inp[0, 0])
            return
        yy, xx = torch.meshgrid(torch.arange(ht, device=dev, dtype=torch.float32), torch.arange(wd, device=dev, dtype=torch.float32), indexing="ij")
        coords0 = torch.stack([xx, yy], -1)[None, None]
        f1, f2 = self.fmap[None, [0]], gmap[None, [0]]
        blk = CorrBlock(f1, f2) if CorrBlock.supported(ht, wd) else CorrBlockRef(f1, f2)
        corr = blk(coords0)
        _, delta, weight = self.update(self.net[None], self.inp[None], corr)[:3]
        self.last_delta = delta.norm(dim=-1).mean().item()
        if self.last_delta > self.thresh:
            self.count = 0
            self.net, self.inp = context()
            self.fmap = gmap
            v.append(tstamp, image[0], None, None, depth, intr8, gmap, self.net[0], self.inp[0])
        else:
            self.count += 1


@torch.no_grad()
def fill_poses(update_op, video, tstamps, fmaps, intrinsics=None, iters=6):
    """Poses of M non-keyframes from their feature maps (trajectory_filler.py:42-84 without the image encoder): linear
    interpolation on SE(3) between the bracketing keyframes (log / exp), the frames appended behind the keyframes, two
    edges each (previous and next keyframe -> frame), `iters` motion-only update iterations.  Returns SE3 [M]."""
    N = video.counter.value
    M = len(tstamps)
    dev = video.device
    tt = torch.as_tensor(tstamps, device=dev, dtype=torch.float)
    ts = video.tstamp[:N]
    Ps = SE3(video.poses[:N])
    t0 = torch.as_tensor([int((ts <= t).sum().item()) - 1 for t in tstamps], device=dev)
    t1 = torch.where(t0 < N - 1, t0 + 1, t0)
    dt = ts[t1] - ts[t0] + 1e-3
    dP = Ps[t1] * Ps[t0].inv()
    vel = dP.log() / dt.unsqueeze(-1)
    Gs = SE3.exp(vel * (tt - ts[t0]).unsqueeze(-1)) * Ps[t0]
    video.counter.value += M
    idx = torch.arange(N, N + M, device=dev)
    video.tstamp[idx] = tt
    video.poses[idx] = Gs.data
    video.disps[idx] = 1.0
    video.intrinsics[idx] = video.intrinsics[0].clone() if intrinsics is None else intrinsics
    video.fmaps[idx] = fmaps
    graph = FactorGraph(video, update_op)
    graph.add_factors(t0, idx)
    graph.add_factors(t1, idx)
    for _ in range(iters):
        graph.update(N, N + M, motion_only=True)
    out = SE3(video.poses[N:N + M].clone())
    video.counter.value -= M
    return out


class PoseTrajectoryFiller:
    """poses of the non-keyframes of an image stream (trajectory_filler.py:21-111): batches of 16 frames -> feature encoder ->
    `fill_poses`.  `nets` = droid_amd.encoder.FeatureNets (only its feature encoder is used, like in the reference)."""

    def __init__(self, nets, update_op, video, batch=16):
        self.nets, self.update, self.video, self.batch = nets, update_op, video, batch
        self.count = 0

    @torch.no_grad()
    def _fill(self, tstamps, images, intrinsics):
        from .encoder import normalize_images
        dev = self.video.device
        images = torch.stack(images, 0).to(dev)                                  # [M,cams,3,H,W] uint8 BGR
        intr = torch.stack(intrinsics, 0).to(dev)
        fmaps = self.nets.fnet(normalize_images(images))                          # [M,cams,128,h,w] (no context features needed)
        N, M = self.video.counter.value, len(tstamps)
        self.video.images[N:N + M] = images[:, 0]
        return [fill_poses(self.update, self.video, tstamps, fmaps, intrinsics=intr / 8.0)]

    @torch.no_grad()
    def __call__(self, image_stream):
        """image_stream yields (tstamp, image [cams,3,H,W] uint8, intrinsics [4]); returns SE3 [number of frames]"""
        import lietorch
        pose_list, tstamps, images, intrinsics = [], [], [], []
        for (tstamp, image, intrinsic) in image_stream:
            tstamps.append(tstamp); images.append(image); intrinsics.append(intrinsic)
            if len(tstamps) == self.batch:
                pose_list += self._fill(tstamps, images, intrinsics)
                tstamps, images, intrinsics = [], [], []
        if len(tstamps) > 0:
            pose_list += self._fill(tstamps, images, intrinsics)
        return lietorch.cat(pose_list, 0)
