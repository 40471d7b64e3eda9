#!/usr/bin/env python
"""GPU: the MotionFilter mirror's flow magnitudes per frame for three thresholds (all accepted / all rejected / the test's),
next to the golden's.  usage: python scripts/debug_motion_filter.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import golden_inputs as gi
from droid_amd.depth_video import DepthVideo
from droid_amd.encoder import FeatureNets, empty_state_dict as enc_sd
from droid_amd.policies import MotionFilter
from droid_amd.update import UpdateModule, empty_state_dict as upd_sd
from droid_amd.weights import deterministic_state_dict


class _S:
    def __init__(self, sd): self.sd = sd
    def state_dict(self): return self.sd


sd = {}
for pre, dim in (("fnet", 128), ("cnet", 256)):
    sd.update({pre + "." + k: v for k, v in deterministic_state_dict(_S(enc_sd(dim)), seed=gi.POLICY_SEEDS[pre]).items()})
nets = FeatureNets().load_state_dict(sd)
for scale in (1.0, 2.0):
    upd = UpdateModule().load_state_dict(deterministic_state_dict(_S(upd_sd()), seed=gi.POLICY_SEEDS["update"], scale=scale))
    for th in (0.0, 1e9, gi.MOTION_FILTER_THRESH):
        video = DepthVideo(image_size=list(gi.POLICY_IMAGE), buffer=16, device="cuda:0")
        mf = MotionFilter(nets, upd, video, thresh=th)
        d, c = [], []
        for k, s in enumerate(gi.MOTION_FILTER_SHIFTS):
            mf.track(float(k), gi.policy_image(7, s), intrinsics=torch.tensor(gi.MOTION_FILTER_INTRINSICS))
            c.append(video.counter.value)
            if k:
                d.append(mf.last_delta)
        print("scale %.1f thresh %g: deltas %s counter %s" % (scale, th, np.round(d, 4), c))
