#!/usr/bin/env python
"""FeatureNets.extract_features on one 384x512 frame (fnet + cnet), 20 calls: wall clock per call; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel table (profiles/r06_encoder_kernel_stats.md)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import numpy as np, torch
from droid_amd.encoder import FeatureNets, empty_state_dict as enc_sd
from droid_amd.weights import deterministic_state_dict


class _S:
    def __init__(self, sd): self.sd = sd
    def state_dict(self): return self.sd


sd = {}
for pre, dim, seed in (("fnet", 128, 11), ("cnet", 256, 12)):
    sd.update({pre + "." + k: v for k, v in deterministic_state_dict(_S(enc_sd(dim)), seed=seed).items()})
nets = FeatureNets().load_state_dict(sd)
H, W = 384, 512
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
img = (torch.rand(1, B, 3, H, W, generator=torch.Generator().manual_seed(0)) * 255).byte().cuda()
for _ in range(3):
    nets.extract_features(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    nets.extract_features(img)
torch.cuda.synchronize()
print("encoders: %.3f ms per call of %d frame(s) of %dx%d" % (1e3 * (time.perf_counter() - t0) / 20, B, H, W))
