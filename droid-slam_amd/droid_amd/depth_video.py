"""DepthVideo on MI355X: the state buffers of a DROID-SLAM session and the geometric operations on them.

Host-side mirror of the reference class (droid_slam/depth_video.py:12-225): same attributes (`poses [buf,7]`,
`disps [buf,h/8,w/8]`, `disps_sens`, `disps_up`, `intrinsics [buf,4]`, `fmaps [buf,rig,128,h/8,w/8]` fp16, `nets`, `inps`
fp16, `tstamp`, `images`, `dirty`, `counter`, `ready`), same methods (`append`, `__setitem__`, `__getitem__`, `reproject`,
`distance`, `ba`, `upsample`, `normalize`, `get_lock`).  What differs is underneath:

  reproject   ONE fused kernel (droid_backends.reproject) instead of ~10 lietorch / elementwise launches
              (geom/projective_ops.py:165-198), same thresholds as the Python path;
  distance    droid_backends.frame_distance (the reference's own API);
  ba          droid_backends.ba: no host round trip, the solve stays on the device;
  upsample    droid_backends.cvx_upsample on the channel-last mask the update operator writes.

One process per GPU: the buffers are plain device tensors (the reference shares them between a tracking and a viewer /
backend process with CUDA IPC, depth_video.py:22-38).
"""
import threading

import torch

import droid_backends as db


class _Counter:
    """`multiprocessing.Value`-shaped counter (value + get_lock) without a second process"""

    def __init__(self, v=0):
        self.value = v
        self._lock = threading.RLock()

    def get_lock(self):
        return self._lock


class DepthVideo:
    def __init__(self, image_size=(480, 640), buffer=1024, stereo=False, device="cuda:0"):
        self.counter = _Counter(0)
        self.ready = _Counter(0)
        self.ht = ht = image_size[0]
        self.wd = wd = image_size[1]
        self.device = torch.device(device)
        f = dict(device=self.device, dtype=torch.float)
        # state (depth_video.py:22-30)
        self.tstamp = torch.zeros(buffer, **f)
        self.images = torch.zeros(buffer, 3, ht, wd, device=self.device, dtype=torch.uint8)
        self.dirty = torch.zeros(buffer, device=self.device, dtype=torch.bool)
        self.red = torch.zeros(buffer, device=self.device, dtype=torch.bool)
        self.poses = torch.zeros(buffer, 7, **f)
        self.disps = torch.ones(buffer, ht // 8, wd // 8, **f)
        self.disps_sens = torch.zeros(buffer, ht // 8, wd // 8, **f)
        self.disps_up = torch.zeros(buffer, ht, wd, **f)
        self.intrinsics = torch.zeros(buffer, 4, **f)
        self.stereo = stereo
        c = 2 if stereo else 1
        # features (depth_video.py:35-38)
        self.fmaps = torch.zeros(buffer, c, 128, ht // 8, wd // 8, dtype=torch.half, device=self.device)
        self.nets = torch.zeros(buffer, 128, ht // 8, wd // 8, dtype=torch.half, device=self.device)
        self.inps = torch.zeros(buffer, 128, ht // 8, wd // 8, dtype=torch.half, device=self.device)
        self.poses[:, 6] = 1.0                                   # identity (depth_video.py:41)
        # per-pixel confidence of the sensor depth = the weight of the depth prior in the BA (BASELINE.json configs[4]).  None: the
        # reference's constant 0.05 (src/droid_kernels.cu:1405-1408) through droid_backends.ba; set_depth_confidence() allocates
        # [buffer, ht/8, wd/8] f32 (0.05 where never set) and ba() then goes through droid_backends.ba_ex
        self.disps_conf = None

    def get_lock(self):
        return self.counter.get_lock()

    # ---- item access (depth_video.py:77-137) ----------------------------------------------------------------------
    def _set(self, index, item):
        if isinstance(index, int) and index >= self.counter.value:
            self.counter.value = index + 1
        elif isinstance(index, torch.Tensor) and index.max().item() > self.counter.value:
            self.counter.value = index.max().item() + 1
        self.tstamp[index] = item[0]
        self.images[index] = item[1]
        if item[2] is not None:
            self.poses[index] = item[2]
        if item[3] is not None:
            self.disps[index] = item[3]
        if item[4] is not None:
            depth = item[4][3::8, 3::8].to(self.device)
            self.disps_sens[index] = torch.where(depth > 0, 1.0 / depth, depth)
        if item[5] is not None:
            self.intrinsics[index] = item[5]
        if len(item) > 6:
            self.fmaps[index] = item[6]
        if len(item) > 7:
            self.nets[index] = item[7]
        if len(item) > 8:
            self.inps[index] = item[8]

    def __setitem__(self, index, item):
        with self.get_lock():
            self._set(index, item)

    def __getitem__(self, index):
        with self.get_lock():
            if isinstance(index, int) and index < 0:
                index = self.counter.value + index
            return (self.poses[index], self.disps[index], self.intrinsics[index], self.fmaps[index],
                    self.nets[index], self.inps[index])

    def append(self, *item):
        with self.get_lock():
            self._set(self.counter.value, item)

    # ---- geometry ---------------------------------------------------------------------------------------------------
    def format_indicies(self, ii, jj):
        if not isinstance(ii, torch.Tensor):
            ii = torch.as_tensor(ii)
        if not isinstance(jj, torch.Tensor):
            jj = torch.as_tensor(jj)
        return (ii.to(device=self.device, dtype=torch.long).reshape(-1).contiguous(),
                jj.to(device=self.device, dtype=torch.long).reshape(-1).contiguous())

    def reproject(self, ii, jj):
        """coords [1,E,h,w,2], valid [1,E,h,w,1] of the pixels of frame ii seen from frame jj (depth_video.py:171-179)"""
        ii, jj = self.format_indicies(ii, jj)
        # per-frame intrinsics like projective_transform (projective_ops.py:180,183): the pose filler writes its own rows
        coords, valid = db.reproject(self.poses, self.disps, self.intrinsics, ii, jj)
        return coords[None], valid[None]

    def distance(self, ii=None, jj=None, beta=0.3, bidirectional=True):
        """mean induced-flow frame distance (depth_video.py:181-211)"""
        return_matrix = False
        if ii is None:
            return_matrix = True
            N = self.counter.value
            ii, jj = torch.meshgrid(torch.arange(N), torch.arange(N), indexing="ij")
        ii, jj = self.format_indicies(ii, jj)
        intr = self.intrinsics[0].contiguous()
        if bidirectional:
            poses = self.poses[:self.counter.value].clone()
            d = 0.5 * (db.frame_distance(poses, self.disps, intr, ii, jj, beta) + db.frame_distance(poses, self.disps, intr, jj, ii, beta))
        else:
            d = db.frame_distance(self.poses, self.disps, intr, ii, jj, beta)
        return d.reshape(N, N) if return_matrix else d

    DEPTH_PRIOR_WEIGHT = 0.05                                     # src/droid_kernels.cu:1405

    def set_depth_confidence(self, index, conf):
        """per-pixel weight of the sensor-depth prior of frame(s) `index`: conf [.., ht/8, wd/8] (or full resolution, sampled at
        [3::8, 3::8] like the depth itself, depth_video.py:93-94), >= 0; applies where disps_sens > 0"""
        if self.disps_conf is None:
            self.disps_conf = torch.full_like(self.disps_sens, self.DEPTH_PRIOR_WEIGHT)
        conf = torch.as_tensor(conf, dtype=torch.float, device=self.device)
        if conf.shape[-2:] == (self.ht, self.wd) and (self.ht, self.wd) != tuple(self.disps_conf.shape[-2:]):
            conf = conf[..., 3::8, 3::8]
        self.disps_conf[index] = conf

    def ba(self, target, weight, eta, ii, jj, t0=1, t1=None, itrs=2, lm=1e-4, ep=0.1, motion_only=False):
        """dense bundle adjustment, in place on poses / disps (depth_video.py:213-225)"""
        with self.get_lock():
            if t1 is None:
                t1 = max(ii.max().item(), jj.max().item()) + 1
            if self.disps_conf is None:
                db.ba(self.poses, self.disps, self.intrinsics[0].contiguous(), self.disps_sens, target, weight, eta, ii, jj,
                      t0, t1, itrs, lm, ep, motion_only)
            else:
                db.ba_ex(self.poses, self.disps, self.intrinsics[0].contiguous(), self.disps_sens, self.disps_conf, target, weight, eta,
                         ii, jj, t0, t1, itrs, lm, ep, motion_only)
            self.disps.clamp_(min=0.001)

    def upsample(self, ix, mask):
        """full-resolution depth of frames ix; mask [K,h,w,576] fp16 channel-last as the update operator of this library
        writes it (or the reference's [1,K,576,h,w]) (depth_video.py:155-159)"""
        if mask.dim() == 5:
            mask = mask[0].permute(0, 2, 3, 1)
        self.disps_up[ix] = db.cvx_upsample(self.disps[ix].contiguous(), mask.contiguous().half())

    def normalize(self):
        """unit mean disparity (depth_video.py:161-168)"""
        with self.get_lock():
            n = self.counter.value
            s = self.disps[:n].mean()
            self.disps[:n] /= s
            self.poses[:n, :3] *= s
            self.dirty[:n] = True
