"""Host-side mirrors of the reference's correlation blocks on top of droid_backends.

Same constructor / call interface as ``CorrBlock`` and ``AltCorrBlock`` of the reference
(droid_slam/modules/corr.py:23-60, 89-117) so that factor-graph code written against them runs as is.
Inference only (no autograd): the hot path runs under ``torch.no_grad()`` in the reference as well
(droid_slam/droid.py:62, droid_backend.py:24).

``CorrBlock``     materialised 4-level pyramid in the MI355X layout of csrc/corr_pyramid.hip (8x8 source blocks,
                  x-adjacent displacement pairs, one record of 25.6 MB per edge at 48x64; ANY image size: on a canvas with h % 8 == 0,
                  w in {16, 32, 64}, transposed when only the rows fit, in 64-column strips when neither dimension does);
                  built on the MFMA (droid_backends.corr_pyramid_build), one launch looks up all four levels
                  (corr_pyramid_lookup), optionally fused with the correlation encoder's first layer (lookup_corr0).
``CorrBlockRef``  the reference layout ``[E, h1, w1, h2/2^l, w2/2^l]`` for any image size; lookup =
                  droid_backends.corr_index_forward per level (FactorGraph takes it when CorrBlock.supported(ht, wd) is false).
``AltCorrBlock``  on-the-fly correlation from pooled feature pyramids; lookup = the MFMA alt-correlation kernel per level
                  (droid_backends.altcorr_forward_nhwc_levels; altcorr_forward for the reference's tensor layout).
"""
import torch
import torch.nn.functional as F

import droid_backends


class CorrBlockRef:
    """Reference-layout pyramid (any image size): volume and pooling by droid_backends.corr_volume_build / corr_volume_pool
    (MFMA contraction over the channels for fp16 features), lookups through droid_backends.corr_index_forward."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3, chunk=64):
        self.num_levels = num_levels
        self.radius = radius
        batch, num, dim, ht, wd = fmap1.shape
        assert batch == 1, "CorrBlock mirrors the inference path (batch == 1)"
        self.corr_pyramid = [
            torch.empty(num, ht, wd, ht // 2 ** l, wd // 2 ** l, dtype=fmap1.dtype, device=fmap1.device)
            for l in range(num_levels)]
        f1, f2 = fmap1[0].contiguous(), fmap2[0].contiguous()
        if f1.dtype not in (torch.float16, torch.float32):
            f1, f2 = f1.float(), f2.float()
        for s in range(0, num, chunk):                      # bounded temporaries, volumes written in place
            e = min(num, s + chunk)
            corr = droid_backends.corr_volume_build(f1[s:e], f2[s:e])          # [n,h,w,h,w]
            for l in range(num_levels):
                self.corr_pyramid[l][s:e] = corr
                if l + 1 < num_levels:
                    corr = droid_backends.corr_volume_pool(corr)

    def __call__(self, coords):
        batch, num, ht, wd, _ = coords.shape
        c = coords.permute(0, 1, 4, 2, 3).contiguous().view(batch * num, 2, ht, wd)
        out = []
        for l in range(self.num_levels):
            corr, = droid_backends.corr_index_forward(self.corr_pyramid[l], c / 2 ** l, self.radius)
            out.append(corr.view(batch, num, -1, ht, wd))
        return torch.cat(out, dim=2)

    def cat(self, other):
        for l in range(self.num_levels):
            self.corr_pyramid[l] = torch.cat([self.corr_pyramid[l], other.corr_pyramid[l]], 0)
        return self

    def __getitem__(self, index):
        for l in range(self.num_levels):
            self.corr_pyramid[l] = self.corr_pyramid[l][index]
        return self

    def bytes(self):
        return sum(v.numel() * v.element_size() for v in self.corr_pyramid)


class AltCorrBlock:
    """Reference interface (corr.py:89-117): AltCorrBlock(fmaps [B,N,C,H,W]); block(coords [B,M,H,W,2], ii, jj) ->
    [B,M,4*49,H,W].  fp16 / 128-channel / H,W % 8 == 0 features (what DROID-SLAM stores, depth_video.py:36) are kept
    channel-last and go through the MFMA kernel (droid_backends.altcorr_forward_nhwc); anything else takes the
    reference-layout entry point altcorr_forward."""

    def __init__(self, fmaps, num_levels=4, radius=3):
        self.num_levels = num_levels
        self.radius = radius
        B, N, C, H, W = fmaps.shape
        f = fmaps.reshape(B * N, C, H, W)
        self.pyramid = []
        for l in range(num_levels):
            self.pyramid.append(f.view(B, N, C, H // 2 ** l, W // 2 ** l))
            if l + 1 < num_levels:
                # 2x2 mean of the last two dims (F.avg_pool2d(f, 2, stride=2), corr.py:100), same kernel as the volume pooling
                f = (droid_backends.corr_volume_pool(f.contiguous()) if f.is_cuda and f.dtype in (torch.float16, torch.float32)
                     else F.avg_pool2d(f, 2, stride=2))
        self.mfma = (B == 1 and radius == 3 and C == 128 and fmaps.dtype == torch.float16 and H % 8 == 0 and W % 8 == 0
                     and fmaps.is_cuda)
        if self.mfma:
            self.nhwc = [p[0].permute(0, 2, 3, 1).contiguous() for p in self.pyramid]     # [N,h,w,C] per level

    def __call__(self, coords, ii, jj):
        c = coords.permute(0, 1, 4, 2, 3).contiguous()
        if self.mfma:
            # the four levels of an edge side by side in one tensor, the 2^-level scaling of the coordinates inside the kernel
            return droid_backends.altcorr_forward_nhwc_levels(self.nhwc[0], self.nhwc, c[0], ii.contiguous(), jj.contiguous())[None]
        outs = []
        for l in range(self.num_levels):
            corr, = droid_backends.altcorr_forward(self.pyramid[0], self.pyramid[l].contiguous(), c / 2 ** l,
                                                   ii, jj, self.radius)
            outs.append(corr.flatten(2, 3))
        return torch.stack(outs, dim=2).flatten(2, 3)


class CorrBlock:
    """MI355X-native pyramid: built on the fp16 MFMA in the displacement-skewed, 8x8-block interleaved
    layout of csrc/corr_pyramid.hip and looked up by ONE fused kernel over the 4 levels.  Same interface as
    the reference CorrBlock (corr.py:23-60): CorrBlock(fmap1, fmap2), block(coords), cat, __getitem__.

    The layout needs h % 8 == 0 and w in {16, 32, 64}.  Any other image size up to 64 columns (TUM's 30x40, 40x64, ...) is
    kept on a zero-padded CANVAS of the next such size: the pooled levels are cut at (h >> l) x (w >> l) like avg_pool2d's
    floor (corr.py:36), so a lookup reads exactly the values / zeros it reads from the reference's volumes; coordinates of
    canvas pixels outside the image point far outside every level, and their outputs are cropped away.  An image with more
    than 64 columns but at most 64 rows (41x73 from a 16:9 video, 60x80) is kept TRANSPOSED on such a canvas: the correlation
    of transposed features at transposed coordinates is the transposed volume, the 7x7 window comes out with its axes swapped
    and is swapped back (or, for lookup_corr0, meets weights with the window swapped: UpdateModule.transposed_twin)."""

    # Images with more than 64 columns AND more than 64 rows (round 5; e.g. 72x96 from a 576x768 input): the correlation volume is
    # linear in the target image, and a lookup reads zero outside a volume -- so the image is cut into column STRIPS of at most 64
    # pixels, one pyramid record per (source strip, target strip) pair on h x 64 canvases, and a lookup is the SUM over the target
    # strips of the lookups at coordinates shifted by the strip's origin (a window that straddles a strip border gets its taps from
    # both records; strips start at multiples of 64, so no 2x2 pooling cell straddles one, and floor pooling of the last strip
    # equals floor pooling of the image: (64 k + r) >> l = (64 k >> l) + (r >> l)).  Same kernels, nS x nT launches' worth of work.

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3, out=None):
        """out: optional storage for the pyramid from CorrBlock.arena(...) (fp16 [>= E, record elements]); the block then is
        a view of its first E records.  A 105 GB hipMalloc takes seconds on a cold device (bench.py `ms_pyramid_alloc`), the
        build itself tens of milliseconds: callers that rebuild pyramids (FactorGraph.update_lowmem) keep the storage."""
        assert num_levels == 4 and radius == 3, "the fused pyramid is specialised to 4 levels / radius 3"
        batch, num, dim, ht, wd = fmap1.shape
        assert batch == 1
        self.num_levels, self.radius = num_levels, radius
        self.ht, self.wd = ht, wd
        self.hc, self.wc = self.canvas(ht, wd)
        self.transposed = self.is_transposed(ht, wd)
        self.strips = self.strip_bounds(ht, wd)                   # [(x0, width)] of the column strips, or None
        f1, f2 = fmap1[0].half(), fmap2[0].half()
        if self.strips is not None:
            assert out is None, "the strip form allocates its own records"
            nS = len(self.strips)
            pad = lambda f, x0, ws: F.pad(f[..., x0:x0 + ws], (0, 64 - ws, 0, self.hc - ht))
            recs = []
            for (xs, ws) in self.strips:                          # source strip
                a = pad(f1, xs, ws).contiguous()
                for (xt, wt) in self.strips:                      # target strip: pooled levels cut at (ht >> l) x (wt >> l)
                    recs.append(droid_backends.corr_pyramid_build(a, pad(f2, xt, wt).contiguous(), ht, wt))
            self.records, self.pyramid = recs, None               # record (source strip si, target strip ti) = records[si * nS + ti], [E, elems]
            return
        if self.transposed:
            f1, f2, ht, wd = f1.transpose(-1, -2), f2.transpose(-1, -2), wd, ht
        if (self.hc, self.wc) != (ht, wd):
            pad = (0, self.wc - wd, 0, self.hc - ht)
            self.pyramid = droid_backends.corr_pyramid_build(F.pad(f1, pad).contiguous(), F.pad(f2, pad).contiguous(), ht, wd, out)
        else:
            self.pyramid = droid_backends.corr_pyramid_build(f1.contiguous(), f2.contiguous(), 0, 0, out)

    @classmethod
    def from_frames(cls, fmaps, ii, jj, out=None):
        """The block the reference builds as CorrBlock(fmaps[ii, 0][None], fmaps[jj, c][None]) (factor_graph.py:128-133; c = 1 on the
        stereo self-edges), from the frame buffer itself: fmaps [N, rig, C, h, w] (video.fmaps), ii / jj [E] int64.  The channel-last
        transpose and the pooled levels are computed once per FRAME that occurs in the edge list (droid_backends.corr_pyramid_prepare_frames)
        instead of once per edge, and the build kernel reads them through the edge's frame indices (corr_pyramid_build_indexed): no
        per-edge feature gathers (2 x 3.2 GB at C3), the same records bit for bit.  Image sizes that are kept transposed or in strips
        take the per-edge constructor."""
        N, rig, C, ht, wd = fmaps.shape
        ii = ii.to(fmaps.device, torch.long).reshape(-1); jj = jj.to(fmaps.device, torch.long).reshape(-1)
        c = (ii == jj).long() if rig > 1 else torch.zeros_like(ii)
        if cls.is_transposed(ht, wd) or cls.strip_bounds(ht, wd) is not None or ii.numel() == 0:
            return cls(fmaps[ii, 0][None], fmaps[jj, c][None], out=out)
        self = cls.__new__(cls)
        self.num_levels, self.radius = 4, 3
        self.ht, self.wd = ht, wd
        self.hc, self.wc = cls.canvas(ht, wd)
        self.transposed, self.strips = False, None
        E = ii.numel()
        frames, inv = torch.unique(torch.cat([ii * rig, jj * rig + c]), return_inverse=True)      # only the frames the edges touch
        f = fmaps.reshape(N * rig, C, ht, wd)[frames].half()
        if (self.hc, self.wc) != (ht, wd):
            f = F.pad(f, (0, self.wc - wd, 0, self.hc - ht))
        prep = droid_backends.corr_pyramid_prepare_frames(f.contiguous(), ht, wd)
        self.pyramid = droid_backends.corr_pyramid_build_indexed(prep, inv[:E].contiguous(), inv[E:].contiguous(), self.hc, self.wc, out)
        return self

    @staticmethod
    def arena(num, ht, wd, device):
        """uninitialised storage for the pyramid of `num` edges of an ht x wd image (CorrBlock(..., out=arena))"""
        return torch.empty(num, CorrBlock.bytes_per_edge(ht, wd) // 2, dtype=torch.float16, device=device)

    @staticmethod
    def is_transposed(ht, wd):
        return wd > 64 and ht <= 64

    @staticmethod
    def strip_bounds(ht, wd):
        """[(x0, width)] of the 64-column strips of an image with more than 64 columns and more than 64 rows, else None"""
        if wd <= 64 or ht <= 64:
            return None
        return [(x0, min(64, wd - x0)) for x0 in range(0, wd, 64)]

    @staticmethod
    def canvas(ht, wd):
        """size of the pyramid's canvas for an ht x wd image (of the transposed image if is_transposed(ht, wd); of ONE column strip
        if the image has more than 64 columns and more than 64 rows: strip_bounds)"""
        if CorrBlock.is_transposed(ht, wd):
            ht, wd = wd, ht
        if wd > 64:
            return (ht + 7) // 8 * 8, 64
        return (ht + 7) // 8 * 8, (16 if wd <= 16 else 32 if wd <= 32 else 64)

    @staticmethod
    def supported(ht, wd):
        return ht >= 1 and wd >= 1

    @staticmethod
    def bytes_per_edge(ht, wd):
        """size of one edge's record (csrc/corr_pyramid.hip make_dims: per level and 8x8 source block (h2 + 1 zero row) x w2
        displacement cells x 64 pixels, fp16) = dh_corr_pyramid_bytes(1, canvas of (ht, wd))"""
        strips = CorrBlock.strip_bounds(ht, wd)
        ht, wd = CorrBlock.canvas(ht, wd)
        nblk = (ht // 8) * (wd // 8)
        return sum(nblk * ((ht >> l) + 1) * (wd >> l) * 64 * 2 for l in range(4)) * (len(strips) ** 2 if strips else 1)

    @staticmethod
    def window_spread(coords):
        """How much of the pyramid layout's coalescing a flow keeps: coords [E,h,w,2] (or [1,E,h,w,2]) -> mean over the 8x8 source
        blocks (= waves of the lookup) of (8 + range_y) (10 + range_x) / 80, range = max - min of the integer tap origin
        floor(coords - coords0) inside the block at level 0.  1.0 = every lane of a wave wants the same displacement cells (a
        translation); a reprojection flow sits at 1.1-1.3; independent random coordinates at ~60.  The number of 128-byte lines
        a wave touches -- and with it the lookup's time -- grows like this figure (scripts/lookup_traffic_model.py), while the
        reference layout (CorrBlockRef) costs the same ~6.4x whatever the flow: FactorGraph switches layouts above SPREAD_LIMIT."""
        c = coords.reshape(-1, *coords.shape[-3:])
        E, h, w, _ = c.shape
        h8, w8 = h // 8 * 8, w // 8 * 8
        if E == 0 or h8 == 0 or w8 == 0:
            return 1.0
        yy, xx = torch.meshgrid(torch.arange(h8, device=c.device, dtype=c.dtype), torch.arange(w8, device=c.device, dtype=c.dtype), indexing="ij")
        d = torch.floor(c[:, :h8, :w8] - torch.stack([xx, yy], -1)).clamp(-4.0 * w, 4.0 * w)        # displacement of the tap origin
        d = d.reshape(E, h8 // 8, 8, w8 // 8, 8, 2)
        rng = d.amax(dim=(2, 4)) - d.amin(dim=(2, 4))                                               # [E, h/8, w/8, 2] (x, y)
        return float(((10.0 + rng[..., 0]) * (8.0 + rng[..., 1])).mean().item() / 80.0)

    SPREAD_LIMIT = 6.0        # the reference-layout kernels take ~6.4x the pyramid lookup's time on a coherent flow (bench.py roofline_sensitivity)

    def _coords(self, coords):
        """[1,E,h,w,2] -> [E,hc,wc,2] contiguous (of the transposed image: pixel (x, y) looks at (y', x')); canvas pixels outside
        the image look at (-1e4, -1e4): every tap outside"""
        batch, num, ht, wd, _ = coords.shape
        assert (ht, wd) == (self.ht, self.wd)
        c = coords.reshape(batch * num, ht, wd, 2)
        if self.transposed:
            c, ht, wd = c.transpose(1, 2).flip(-1), wd, ht
        if (self.hc, self.wc) != (ht, wd):
            c = F.pad(c, (0, 0, 0, self.wc - wd, 0, self.hc - ht), value=-1.0e4)
        return c.contiguous()

    def _strip_lookup(self, coords):
        """[1,E,h,w,2] -> [E,196,h,w] fp16: per source strip the sum over the target strips of the record's lookup at shifted coordinates"""
        batch, num, ht, wd, _ = coords.shape
        c = coords.reshape(batch * num, ht, wd, 2)
        nS = len(self.strips)
        cols = []
        for si, (xs, ws) in enumerate(self.strips):
            cs = F.pad(c[:, :, xs:xs + ws], (0, 0, 0, 64 - ws, 0, self.hc - ht), value=-1.0e4)            # canvas pixels look far outside
            acc = None
            for ti, (xt, wt) in enumerate(self.strips):
                shift = torch.tensor([float(xt), 0.0], device=c.device)
                o = droid_backends.corr_pyramid_lookup(self.records[si * nS + ti], (cs - shift).contiguous())[:, :, :ht, :ws]
                acc = o.float() if acc is None else acc + o.float()
            cols.append(acc.half())
        return torch.cat(cols, -1)

    def __call__(self, coords):
        batch, num, ht, wd, _ = coords.shape
        if self.strips is not None:
            return self._strip_lookup(coords).reshape(batch, num, -1, ht, wd)
        out = droid_backends.corr_pyramid_lookup(self.pyramid, self._coords(coords))
        if self.transposed:           # [E, level, xoff' = yoff, yoff' = xoff, x, y] -> [E, level, xoff, yoff, y, x]
            out = out[:, :, :wd, :ht].reshape(batch * num, 4, 7, 7, wd, ht).permute(0, 1, 3, 2, 5, 4)
            return out.reshape(batch, num, -1, ht, wd)
        return out[:, :, :ht, :wd].reshape(batch, num, -1, ht, wd)

    def lookup_nhwc(self, coords):
        """[1,E,h,w,2] -> [4,E,h,w,56] level-planar channel-last features for droid_amd.update.UpdateModule.forward_nhwc"""
        batch, num, ht, wd, _ = coords.shape
        if self.strips is not None:                          # reference-layout sum -> the level-planar channel-last form
            feats = self._strip_lookup(coords).view(num, 4, 7, 7, ht, wd)                   # [E, level, xoff, yoff, h, w]
            out = torch.zeros(4, num, ht, wd, 56, dtype=feats.dtype, device=feats.device)
            out[..., :49] = feats.permute(1, 0, 4, 5, 3, 2).reshape(4, num, ht, wd, 49)      # channel = yoff * 7 + xoff
            return out
        out = droid_backends.corr_pyramid_lookup_nhwc(self.pyramid, self._coords(coords))
        if self.transposed:           # channel = yoff' * 7 + xoff' = xoff * 7 + yoff  ->  yoff * 7 + xoff
            k = torch.arange(56, device=out.device)
            perm = torch.where(k < 49, (k % 7) * 7 + k // 7, k)
            return out[:, :, :wd, :ht].transpose(2, 3)[..., perm].contiguous()
        return out if (self.hc, self.wc) == (ht, wd) else out[:, :, :ht, :wd].contiguous()

    def lookup_corr0(self, coords, update_op):
        """[1,E,h,w,2] -> [E,h,w,128] f16: the lookup and the first layer of the update operator's correlation encoder
        (Conv2d(196,128,1) + ReLU, droid_net.py:96-100) in one kernel; pass it to UpdateModule.forward_nhwc(corr0=...)"""
        batch, num, ht, wd, _ = coords.shape
        if self.strips is not None:                          # ReLU(layer) does not distribute over the strips' sum: lookup, then the layer
            return update_op.corr0_layer(self._strip_lookup(coords))
        wpk, bias = (update_op.transposed_twin() if self.transposed else update_op).params["corr0_fused"]
        out = droid_backends.corr_pyramid_lookup_corr0(self.pyramid, self._coords(coords), wpk, bias)
        if self.transposed:
            return out[:, :wd, :ht].transpose(1, 2).contiguous()
        return out if (self.hc, self.wc) == (ht, wd) else out[:, :ht, :wd].contiguous()

    def cat(self, other):
        if self.strips is not None:
            self.records = [torch.cat([a, b], 0) for a, b in zip(self.records, other.records)]
            return self
        self.pyramid = torch.cat([self.pyramid, other.pyramid], 0)
        return self

    def __getitem__(self, index):
        if self.strips is not None:
            self.records = [r[index] for r in self.records]
            return self
        self.pyramid = self.pyramid[index]
        return self

    def bytes(self):
        if self.strips is not None:
            return sum(r.numel() * r.element_size() for r in self.records)
        return self.pyramid.numel() * self.pyramid.element_size()
