"""CPU stand-in for the un-vendored ``torch_scatter`` (reference .gitmodules:4-6) -- GOLDEN GENERATION ONLY."""
import torch


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    dim = dim % src.dim()
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    shape = list(src.shape); shape[dim] = dim_size
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    return res.index_add_(dim, index.to(src.device), src)


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    dim = dim % src.dim()
    s = scatter_sum(src, index, dim, dim_size=dim_size)
    cnt = torch.zeros(s.shape[dim], dtype=src.dtype, device=src.device)
    cnt.index_add_(0, index.to(src.device), torch.ones(index.shape[0], dtype=src.dtype, device=src.device))
    shape = [1] * s.dim(); shape[dim] = -1
    return s / cnt.clamp(min=1).view(shape)
