"""Reconstruction output of a session: the on-disk dump and the filtered point cloud the viewers draw.

Mirrors of the reference's consumers of `droid_backends.iproj` / `droid_backends.depth_filter`:
  save_reconstruction / load_reconstruction   demo.py:60-76 (same keys, same tensors: a torch.save'd dict)
  point_cloud                                 view_reconstruction.py:15-38 and visualization.py:97-114 (back-project every
                                              pixel with the camera-to-world pose, keep points seen consistently by
                                              >= `filter_count` neighbouring frames and not too far away)
No viewer here (Open3D / moderngl are outside the path): the functions return tensors.
"""
import torch

import droid_backends as db
from lietorch import SE3


def save_reconstruction(video, save_path):
    """{"tstamps", "images", "disps" (full resolution, disps_up), "poses", "intrinsics"} of the first `counter` frames"""
    t = video.counter.value
    torch.save({"tstamps": video.tstamp[:t].cpu(), "images": video.images[:t].cpu(), "disps": video.disps_up[:t].cpu(),
                "poses": video.poses[:t].cpu(), "intrinsics": video.intrinsics[:t].cpu()}, save_path)


def load_reconstruction(path, device="cuda"):
    blob = torch.load(path, map_location="cpu")
    return {k: v.to(device) for k, v in blob.items()}


@torch.no_grad()
def point_cloud(poses, disps, intrinsics, images=None, filter_thresh=0.005, filter_count=2, min_disp_frac=0.25):
    """poses [N,7] world->camera, disps [N,h,w], intrinsics [4] at the resolution of disps, images [N,3,h,w] uint8 (BGR)
    -> (points [M,3] world coordinates, colors [M,3] RGB in [0,1] or None, mask [N,h,w])."""
    poses = poses.contiguous().float(); disps = disps.contiguous().float(); intrinsics = intrinsics.contiguous().float()
    index = torch.arange(poses.shape[0], device=poses.device)
    thresh = filter_thresh * torch.ones_like(disps.mean(dim=[1, 2]))
    points = db.iproj(SE3(poses).inv().data.contiguous(), disps, intrinsics)
    counts = db.depth_filter(poses, disps, intrinsics, index, thresh)
    mask = (counts >= filter_count) & (disps > min_disp_frac * disps.mean())
    colors = None
    if images is not None:
        colors = (images[:, [2, 1, 0]].permute(0, 2, 3, 1).float() / 255.0)[mask]
    return points[mask], colors, mask
