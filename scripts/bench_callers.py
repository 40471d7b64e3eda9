#!/usr/bin/env python
"""Timings of the path's callers on one MI355X, synthetic data, random-init weights:
  encoders        FeatureNets.extract_features on one 384x512 frame (fnet + cnet, what MotionFilter runs per incoming frame)
  motion filter   MotionFilter.track per frame (encoders + one-edge pyramid + one update iteration)
  frontend        DroidFrontend: initialisation (8 + 8 update iterations on 8 keyframes) and one keyframe update
  proximity       FactorGraph.add_proximity_factors over 512 keyframes (512 x 512 frame distances + device NMS)
  pose filler     PoseTrajectoryFiller on 16 non-keyframes (encoder + 6 motion-only iterations)
usage: python scripts/bench_callers.py"""
import json, os, sys, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import numpy as np, torch
from droid_amd import synthetic as syn
from droid_amd.depth_video import DepthVideo
from droid_amd.encoder import FeatureNets, empty_state_dict as enc_sd
from droid_amd.factor_graph import FactorGraph
from droid_amd.policies import MotionFilter, DroidFrontend, PoseTrajectoryFiller
from droid_amd.update import UpdateModule, empty_state_dict as upd_sd
from droid_amd.weights import deterministic_state_dict


class _S:
    def __init__(self, sd): self.sd = sd
    def state_dict(self): return self.sd


def wall(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


sd = {}
for pre, dim, seed in (("fnet", 128, 11), ("cnet", 256, 12)):
    sd.update({pre + "." + k: v for k, v in deterministic_state_dict(_S(enc_sd(dim)), seed=seed).items()})
nets = FeatureNets().load_state_dict(sd)
upd = UpdateModule().load_state_dict(deterministic_state_dict(_S(upd_sd()), seed=1234))
out = {}
H, W = 384, 512
g = torch.Generator().manual_seed(0)
img = (torch.rand(1, 3, H, W, generator=g) * 255).byte()
# ---- encoders
x = img[None].cuda()
out["encoders_ms_per_frame_384x512"] = wall(lambda: nets.extract_features(x))
# ---- motion filter
video = DepthVideo(image_size=[H, W], buffer=64, device="cuda:0")
mf = MotionFilter(nets, upd, video, thresh=1e9)             # never accepts after the first frame: steady-state cost of a rejected frame
intr = torch.tensor([320.0, 320.0, 256.0, 192.0])
mf.track(0.0, img, intrinsics=intr)
out["motion_filter_ms_per_frame"] = wall(lambda: mf.track(1.0, img, intrinsics=intr))
# ---- frontend on synthetic keyframes (48 x 64 features)
gC = syn.make_graph(syn.GraphConfig("fe", 16, 60, radius=2), with_features=True)
N = 16


def fresh_video(n_present):
    v = DepthVideo(image_size=[H, W], buffer=N + 24, device="cuda:0")
    d = lambda a: torch.as_tensor(a).cuda()
    v.poses[:N] = d(gC["poses"]); v.disps[:N] = d(gC["disps"]); v.intrinsics[:N] = d(gC["intrinsics"])
    v.fmaps[:N] = d(gC["fmaps"]); v.nets[:N] = d(gC["nets"]); v.inps[:N] = d(gC["inps"])
    v.tstamp[:N] = torch.arange(N, device="cuda").float()
    v.counter.value = n_present
    return v


args = SimpleNamespace(upsample=True, warmup=8, beta=0.3, frontend_nms=1, keyframe_thresh=0.0, frontend_window=20, frontend_thresh=16.0, frontend_radius=2)


def fe_init():
    v = fresh_video(8)
    fe = DroidFrontend(upd, v, args)
    fe()
    return v, fe


out["frontend_init_ms_8kf_16_iterations"] = wall(lambda: fe_init(), reps=2)
v, fe = fe_init()


def fe_step():
    v.counter.value = fe.t1 + 1
    fe()


fe_step(); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while fe.t1 < N - 1:
    fe_step(); n += 1
torch.cuda.synchronize()
out["frontend_update_ms_per_keyframe"] = 1e3 * (time.perf_counter() - t0) / max(1, n)
out["frontend_edges_at_the_end"] = int(len(fe.graph.ii))
# ---- proximity factors over a 512-keyframe video
g3 = syn.make_graph("C3")
v3 = DepthVideo(image_size=[H, W], buffer=520, device="cuda:0")
v3.poses[:512] = torch.as_tensor(g3["poses_gt"]).cuda(); v3.disps[:512] = torch.as_tensor(g3["disps_gt"]).cuda()
v3.intrinsics[:512] = torch.as_tensor(g3["intrinsics"]).cuda(); v3.counter.value = 512


def prox():
    fg = FactorGraph(v3, upd, corr_impl="alt", max_factors=16 * 512)
    fg.add_proximity_factors(rad=2, nms=2, thresh=16.0, beta=0.25)
    return fg


out["proximity_factors_ms_512kf"] = wall(prox, reps=2)
out["proximity_edges_512kf"] = int(len(prox().ii))
# ---- pose filler: 16 frames
vf = fresh_video(16)
stream = [(float(k) + 0.5, img, intr) for k in range(16)]
filler = PoseTrajectoryFiller(nets, upd, vf)
out["pose_filler_ms_per_16_frames"] = wall(lambda: filler(stream), reps=2)
print(json.dumps(out))
