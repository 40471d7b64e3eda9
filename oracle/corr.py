"""Oracle restatement of the correlation volume build and the two lookup kernels.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  corr_volume / corr_pyramid ... droid_slam/modules/corr.py:23-38, 63-71
  corr_index_forward ............ src/correlation_kernels.cu:20-71, 127-156
  corr_index_backward ........... src/correlation_kernels.cu:74-125, 158-186
  corr_block_lookup ............. droid_slam/modules/corr.py:40-50   (CorrBlock.__call__)
  altcorr_forward ............... src/altcorr_kernel.cu:24-75, 132-172
  altcorr_backward .............. src/altcorr_kernel.cu:78-129, 175-225
  alt_block_lookup .............. droid_slam/modules/corr.py:89-117  (AltCorrBlock)

Everything is computed in float64 from the (possibly fp16) inputs; the reference
accumulates the volume lookup in the volume dtype in global memory and rounds
each feature product of the alt path to the feature dtype -- those rounding
orders are NOT reproduced (documented tolerance, SURVEY.md section 7 "fp16 semantics").
"""
import numpy as np


def corr_volume(fmap1, fmap2):
    """All-pairs correlation.  fmap* [E,C,h,w] -> [E,h,w,h,w];  (f1/4)^T (f2/4)."""
    E, C, h, w = fmap1.shape
    a = fmap1.reshape(E, C, h * w).astype(np.float64) / 4.0
    b = fmap2.reshape(E, C, h * w).astype(np.float64) / 4.0
    return np.einsum("ecp,ecq->epq", a, b).reshape(E, h, w, h, w)


def avg_pool2(x):
    """2x2 average pooling over the last two axes (F.avg_pool2d(.,2,stride=2), floor mode)."""
    h, w = x.shape[-2] // 2 * 2, x.shape[-1] // 2 * 2
    x = x[..., :h, :w]
    return 0.25 * (x[..., 0::2, 0::2] + x[..., 0::2, 1::2] + x[..., 1::2, 0::2] + x[..., 1::2, 1::2])


def corr_pyramid(fmap1, fmap2, num_levels=4):
    vol = corr_volume(fmap1, fmap2)
    pyr = []
    for _ in range(num_levels):
        pyr.append(vol)
        vol = avg_pool2(vol)
    return pyr


def corr_index_forward(volume, coords, radius):
    """volume [N,h1,w1,h2,w2], coords [N,2,h1,w1] (x,y) -> corr [N,2r+1,2r+1,h1,w1] (float64).

    Output index order is (x-offset, y-offset), as in the kernel (:47-67)."""
    N, h1, w1, h2, w2 = volume.shape
    r = radius
    rd = 2 * r + 1
    vol = volume.astype(np.float64)
    x0 = coords[:, 0].astype(np.float32)
    y0 = coords[:, 1].astype(np.float32)
    fx = np.floor(x0)
    fy = np.floor(y0)
    dx = (x0 - fx).astype(np.float64)
    dy = (y0 - fy).astype(np.float64)
    fxi = fx.astype(np.int64)
    fyi = fy.astype(np.int64)
    out = np.zeros((N, rd, rd, h1, w1), dtype=np.float64)
    n_ix, y_ix, x_ix = np.meshgrid(np.arange(N), np.arange(h1), np.arange(w1), indexing="ij")
    for i in range(rd + 1):
        for j in range(rd + 1):
            x1 = fxi - r + i
            y1 = fyi - r + j
            ok = (x1 >= 0) & (x1 < w2) & (y1 >= 0) & (y1 < h2)
            s = np.where(ok, vol[n_ix, y_ix, x_ix, np.clip(y1, 0, h2 - 1), np.clip(x1, 0, w2 - 1)], 0.0)
            if i > 0 and j > 0:
                out[:, i - 1, j - 1] += s * (dx * dy)
            if i > 0 and j < rd:
                out[:, i - 1, j] += s * (dx * (1 - dy))
            if i < rd and j > 0:
                out[:, i, j - 1] += s * ((1 - dx) * dy)
            if i < rd and j < rd:
                out[:, i, j] += s * ((1 - dx) * (1 - dy))
    return out


def corr_index_backward(volume_shape, coords, corr_grad, radius):
    """Adjoint of corr_index_forward w.r.t. the volume (:74-125)."""
    N, h1, w1, h2, w2 = volume_shape
    r = radius
    rd = 2 * r + 1
    g = corr_grad.astype(np.float64)
    x0 = coords[:, 0].astype(np.float32)
    y0 = coords[:, 1].astype(np.float32)
    fx = np.floor(x0); fy = np.floor(y0)
    dx = (x0 - fx).astype(np.float64); dy = (y0 - fy).astype(np.float64)
    fxi = fx.astype(np.int64); fyi = fy.astype(np.int64)
    vg = np.zeros(volume_shape, dtype=np.float64)
    n_ix, y_ix, x_ix = np.meshgrid(np.arange(N), np.arange(h1), np.arange(w1), indexing="ij")
    for i in range(rd + 1):
        for j in range(rd + 1):
            x1 = fxi - r + i
            y1 = fyi - r + j
            ok = (x1 >= 0) & (x1 < w2) & (y1 >= 0) & (y1 < h2)
            acc = np.zeros((N, h1, w1))
            if i > 0 and j > 0:
                acc += g[:, i - 1, j - 1] * (dx * dy)
            if i > 0 and j < rd:
                acc += g[:, i - 1, j] * (dx * (1 - dy))
            if i < rd and j > 0:
                acc += g[:, i, j - 1] * ((1 - dx) * dy)
            if i < rd and j < rd:
                acc += g[:, i, j] * ((1 - dx) * (1 - dy))
            np.add.at(vg, (n_ix[ok], y_ix[ok], x_ix[ok], y1[ok], x1[ok]), acc[ok])
    return vg


def corr_block_lookup(pyramid, coords, radius=3):
    """CorrBlock.__call__: coords [E,h,w,2] -> [E, L*(2r+1)^2, h, w], channel = l*49 + a*7 + b."""
    E, h, w, _ = coords.shape
    c = np.ascontiguousarray(np.moveaxis(coords, -1, 1)).astype(np.float32)   # [E,2,h,w]
    outs = []
    for l, vol in enumerate(pyramid):
        o = corr_index_forward(vol, (c / np.float32(2 ** l)).astype(np.float32), radius)
        outs.append(o.reshape(E, -1, h, w))
    return np.concatenate(outs, axis=1)


def corr_block_lookup_torch(pyramid, coords, radius=3):
    """The same lookup through torch's multi-threaded grid_sample (the kernel is equivalent to
    F.grid_sample(align_corners=True, padding_mode="zeros") on each pixel's own slice): the
    multi-core form used by bench.py's cpu_baseline.  pyramid: list of torch tensors [E,h,w,h2,w2]; coords [E,h,w,2]."""
    import torch
    import torch.nn.functional as F
    E, h, w, _ = coords.shape
    r = radius
    dd = torch.arange(-r, r + 1, dtype=torch.float32)
    outs = []
    for l, vol in enumerate(pyramid):
        h2, w2 = vol.shape[-2:]
        c = coords.reshape(E * h * w, 1, 1, 2) / float(2 ** l)
        gx = c[..., 0] + dd.view(1, -1, 1)                     # [N, a (x offset), 1]
        gy = c[..., 1] + dd.view(1, 1, -1)                     # [N, 1, b (y offset)]
        gx, gy = torch.broadcast_tensors(gx, gy)               # [N, a, b]
        grid = torch.stack([2 * gx / max(w2 - 1, 1) - 1, 2 * gy / max(h2 - 1, 1) - 1], -1)
        o = F.grid_sample(vol.reshape(E * h * w, 1, h2, w2).float(), grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        outs.append(o.view(E, h, w, (2 * r + 1) ** 2).permute(0, 3, 1, 2))    # channel = a*7 + b
    return torch.cat(outs, 1)


def altcorr_forward(fmap1, fmap2, coords, ii, jj, radius):
    """fmap1 [B,N,C,H,W], fmap2 [B,N,C,H2,W2], coords [B,M,2,H,W] -> [B,M,2r+1,2r+1,H,W] (x-offset outer).

    Integer-tap dot products (:24-75) + bilinear blend (:160-169) + permute (:171)."""
    B, M = coords.shape[:2]
    H, W = coords.shape[3:]
    C = fmap1.shape[2]
    H2, W2 = fmap2.shape[3:]
    R = radius
    D = 2 * R + 2
    f1 = fmap1.astype(np.float64) / 4.0
    f2 = fmap2.astype(np.float64) / 4.0
    out = np.zeros((B, M, D - 1, D - 1, H, W), dtype=np.float64)
    for b in range(B):
        for m in range(M):
            a = f1[b, int(ii[m])]                      # [C,H,W]
            t = f2[b, int(jj[m])]                      # [C,H2,W2]
            x = coords[b, m, 0].astype(np.float32)
            y = coords[b, m, 1].astype(np.float32)
            fxq = np.floor(x); fyq = np.floor(y)
            dx = (x - fxq).astype(np.float64); dy = (y - fyq).astype(np.float64)
            corr = np.zeros((D, D, H, W))              # [ii (y tap), jj (x tap)]
            for ti in range(D):
                for tj in range(D):
                    i1 = fyq.astype(np.int64) + (ti - R)
                    j1 = fxq.astype(np.int64) + (tj - R)
                    ok = (i1 >= 0) & (i1 < H2) & (j1 >= 0) & (j1 < W2)
                    g = t[:, np.clip(i1, 0, H2 - 1), np.clip(j1, 0, W2 - 1)]   # [C,H,W]
                    corr[ti, tj] = np.where(ok, np.sum(a * g, axis=0), 0.0)
            o = ((1 - dx) * (1 - dy) * corr[:D - 1, :D - 1] + dx * (1 - dy) * corr[:D - 1, 1:]
                 + (1 - dx) * dy * corr[1:, :D - 1] + dx * dy * corr[1:, 1:])    # [y-off, x-off]
            out[b, m] = np.transpose(o, (1, 0, 2, 3))                             # -> [x-off, y-off]
    return out


def altcorr_forward_fast(fmap1, fmap2, coords, ii, jj, radius, edge_chunk=8):
    """altcorr_forward for LARGE edge counts (the C5-size golden of tests/golden/make_graph_scale_golden.py: 8192 edges x 4 levels x 2
    steps): the same integer-tap dot products (altcorr_kernel.cu:24-75), taken out of the edge's all-pairs product (f1/4)^T (f2/4)
    (one GEMM per edge) instead of 64 gathers of the 128-channel target map; same bounds rule (a tap outside the target image is 0),
    same bilinear blend (:160-169), same [x-off, y-off] output (:171).  torch CPU, fp64 accumulation of fp32 products' sums differs
    from altcorr_forward by rounding only: pinned to it by tests/test_oracle_golden.py::test_fast_altcorr_equals_altcorr."""
    import torch
    B, M = coords.shape[:2]
    H, W = coords.shape[3:]
    H2, W2 = fmap2.shape[3:]
    R = radius
    D = 2 * R + 2
    # (torch tensors are taken as they are and converted per edge chunk: the C5 feature buffer is 3.2 GB in fp32)
    f1 = fmap1 if isinstance(fmap1, torch.Tensor) else torch.as_tensor(np.asarray(fmap1))
    f2 = fmap2 if isinstance(fmap2, torch.Tensor) else torch.as_tensor(np.asarray(fmap2))
    coords = coords.numpy() if isinstance(coords, torch.Tensor) else coords
    C = f1.shape[2]
    out = np.zeros((B, M, D - 1, D - 1, H, W), dtype=np.float64)
    off = torch.arange(D) - R
    for b in range(B):
        for s in range(0, M, edge_chunk):
            e = slice(s, min(M, s + edge_chunk))
            n = e.stop - e.start
            a = f1[b, torch.as_tensor(np.asarray(ii[e], dtype=np.int64))].float().reshape(n, C, H * W) / 4.0          # [n,C,HW]
            t = f2[b, torch.as_tensor(np.asarray(jj[e], dtype=np.int64))].float().reshape(n, C, H2 * W2) / 4.0
            vol = torch.bmm(a.transpose(1, 2).double(), t.double())                                     # [n,HW,H2W2] fp64
            x = torch.as_tensor(np.asarray(coords[b, e, 0], dtype=np.float32)).reshape(n, H * W)
            y = torch.as_tensor(np.asarray(coords[b, e, 1], dtype=np.float32)).reshape(n, H * W)
            fx, fy = torch.floor(x), torch.floor(y)
            dx, dy = (x - fx).double(), (y - fy).double()
            i1 = fy.long()[:, :, None, None] + off[None, None, :, None]                                 # [n,HW,D(y tap),1]
            j1 = fx.long()[:, :, None, None] + off[None, None, None, :]                                 # [n,HW,1,D(x tap)]
            ok = (i1 >= 0) & (i1 < H2) & (j1 >= 0) & (j1 < W2)
            idx = (i1.clamp(0, H2 - 1) * W2 + j1.clamp(0, W2 - 1)).reshape(n, H * W, D * D)
            taps = torch.gather(vol, 2, idx).reshape(n, H * W, D, D) * ok
            dx_, dy_ = dx[:, :, None, None], dy[:, :, None, None]
            o = ((1 - dx_) * (1 - dy_) * taps[:, :, :D - 1, :D - 1] + dx_ * (1 - dy_) * taps[:, :, :D - 1, 1:]
                 + (1 - dx_) * dy_ * taps[:, :, 1:, :D - 1] + dx_ * dy_ * taps[:, :, 1:, 1:])            # [n,HW,y-off,x-off]
            out[b, e] = o.permute(0, 3, 2, 1).reshape(n, D - 1, D - 1, H, W).numpy()                   # -> [x-off, y-off, H, W]
    return out


def altcorr_backward(fmap1, fmap2, coords, corr_grad, ii, jj, radius):
    """Gradients of the alt lookup w.r.t. both feature maps (altcorr_kernel.cu:78-129, 175-225).

    corr_grad [B,M,2r+1,2r+1,H,W] is given in the forward's OUTPUT layout (x-offset outer); the wrapper permutes it
    back (:189), spreads it onto the (2r+2)^2 integer taps with the bilinear weights (:196-205) and the kernel
    scatters g*fmap2 / g*fmap1 (:117-125).  QUIRK, pinned by the reference-produced golden vectors: the kernel
    multiplies with the RAW features, i.e. the backward is NOT scaled by the 1/16 the forward applies (f1/4 * f2/4),
    so it is 16x the true adjoint.  Returns (fmap1_grad, fmap2_grad) in float64."""
    B, M = coords.shape[:2]
    H, W = coords.shape[3:]
    H2, W2 = fmap2.shape[3:]
    R = radius
    D = 2 * R + 2
    f1 = fmap1.astype(np.float64)
    f2 = fmap2.astype(np.float64)
    g1 = np.zeros_like(f1)
    g2 = np.zeros_like(f2)
    for b in range(B):
        for m in range(M):
            ix, jx = int(ii[m]), int(jj[m])
            x = coords[b, m, 0].astype(np.float32)
            y = coords[b, m, 1].astype(np.float32)
            fxq = np.floor(x); fyq = np.floor(y)
            dx = (x - fxq).astype(np.float64); dy = (y - fyq).astype(np.float64)
            g = np.transpose(corr_grad[b, m].astype(np.float64), (1, 0, 2, 3))      # -> [y-off, x-off, H, W]
            tap = np.zeros((D, D, H, W))
            tap[:D - 1, :D - 1] += (1 - dx) * (1 - dy) * g
            tap[:D - 1, 1:] += dx * (1 - dy) * g
            tap[1:, :D - 1] += (1 - dx) * dy * g
            tap[1:, 1:] += dx * dy * g
            yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
            for ti in range(D):
                for tj in range(D):
                    i1 = fyq.astype(np.int64) + (ti - R)
                    j1 = fxq.astype(np.int64) + (tj - R)
                    ok = (i1 >= 0) & (i1 < H2) & (j1 >= 0) & (j1 < W2)
                    if not ok.any():
                        continue
                    gv = tap[ti, tj][ok]                                              # [n]
                    a = f1[b, ix][:, yy[ok], xx[ok]]                                  # [C,n]
                    t = f2[b, jx][:, i1[ok], j1[ok]]                                  # [C,n]
                    np.add.at(g1[b, ix], (slice(None), yy[ok], xx[ok]), gv * t)
                    np.add.at(g2[b, jx], (slice(None), i1[ok], j1[ok]), gv * a)
    return g1, g2


def feature_pyramid(fmaps, num_levels=4):
    """AltCorrBlock.__init__: fmaps [B,N,C,H,W] -> list of avg-pooled levels."""
    pyr = []
    f = fmaps.astype(np.float64)
    for _ in range(num_levels):
        pyr.append(f)
        f = avg_pool2(f)
    return pyr


def alt_block_lookup(fmaps, coords, ii, jj, radius=3, num_levels=4, pool_dtype=None):
    """AltCorrBlock.__call__: coords [B,M,H,W,2] -> [B,M,L*49,H,W].

    ``pool_dtype`` (e.g. np.float16) rounds each pooled pyramid level like the
    reference, whose pyramid is stored in the feature dtype."""
    pyr = feature_pyramid(fmaps, num_levels)
    if pool_dtype is not None:
        pyr = [p.astype(pool_dtype).astype(np.float64) for p in pyr]
    c = np.ascontiguousarray(np.moveaxis(coords, -1, 2)).astype(np.float32)
    outs = []
    for l in range(num_levels):
        o = altcorr_forward(pyr[0], pyr[l], (c / np.float32(2 ** l)).astype(np.float32), ii, jj, radius)
        outs.append(o.reshape(o.shape[0], o.shape[1], -1, o.shape[4], o.shape[5]))
    return np.concatenate(outs, axis=2)
