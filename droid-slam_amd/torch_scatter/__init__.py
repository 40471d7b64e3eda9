"""`torch_scatter` drop-in for the two call sites on DROID-SLAM's BA update path (reference geom/ba.py:8-28
scatter_sum, droid_net.py:18,67 scatter_mean; the un-vendored rusty1s/pytorch_scatter in the reference).
Plain index_add_ on the tensor's own device; the fused path of this repository does not go through here
(droid_backends.segment_mean / the frame-centric BA build do the same reductions inside the HIP kernels)."""
import torch


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    dim = dim % src.dim()
    if index.dim() != 1:
        raise NotImplementedError("only 1-D indices along `dim` are used on the BA update path")
    n = int(index.max().item()) + 1 if dim_size is None and index.numel() else (dim_size or 0)
    shape = list(src.shape); shape[dim] = n
    res = torch.zeros(shape, dtype=src.dtype, device=src.device) if out is None else out
    return res.index_add_(dim, index, src)


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    dim = dim % src.dim()
    s = scatter_sum(src.float(), index, dim, None, dim_size)
    cnt = torch.zeros(s.shape[dim], dtype=torch.float32, device=src.device).index_add_(
        0, index, torch.ones(index.numel(), dtype=torch.float32, device=src.device)).clamp_(min=1)
    view = [1] * s.dim(); view[dim] = -1
    return (s / cnt.view(view)).to(src.dtype)
