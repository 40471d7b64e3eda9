"""FactorGraph partitioned by edge batches across the GPUs of one node (BASELINE.json configs[3] / configs[4]).

north_star: "global BA over the full frame graph is partitioned by edge batches across the 8 GPUs of one node with RCCL
all-reduce of the 6x6 pose Hessian blocks and residuals".  The global BA of the reference is DroidBackend.__call__ ->
FactorGraph.update_lowmem (droid_slam/droid_backend.py:25-42, factor_graph.py:266-330); the reference runs it on one device.
`DistFactorGraph` is that class -- same constructor, attributes and methods as droid_amd.factor_graph.FactorGraph, i.e. as the
reference's -- with one process per GPU:

  replicated on every rank   the DepthVideo (poses, depths, feature / context maps: per-FRAME data, 0.4 GB of features at 512
                             keyframes) and the edge LISTS `ii, jj, age, ii_inac, jj_inac, ii_bad, jj_bad` (a few KB; every
                             decision about them -- proximity edges, NMS, ageing, max_factors -- is a function of replicated
                             data, so all ranks take it identically without talking);
  sharded by SOURCE frame    everything that is per EDGE and large: the correlation pyramid records (25.6 MB per edge), the
                             hidden state, targets, weights, their inactive copies.  Rank r holds the edges whose source frame lies
                             in [bounds[r], bounds[r+1]) -- contiguous frame ranges balanced by edge count
                             (dist_ba.shard_edges_by_source_frame, the axis the reference chunks on, factor_graph.py:284-287) --
                             in the order of the global list.  GraphAgg's mean over the edges of a source frame and the depth
                             block of a frame never leave the rank.

An update iteration is therefore rank-local up to the BA: reproject -> pyramid lookup -> update operator -> ba_inputs have no
collective; `_solve` hands this rank's targets / weights to dist_ba.DistBA.ba (per Gauss-Newton iteration ONE all-reduce of the
co-visible 6x6 blocks + rhs; at the end one all-reduce of the depth maps), after which every rank holds identical poses / disps.
`update_lowmem` builds the pyramid of the rank's edges once per call (CorrBlock.from_frames into the rank's arena: 105 GB / world
at C3).  With `upsample`, the full-resolution depth maps a rank computed for its frames are exchanged once per call.

The frame ranges are fixed by the first edges added to an empty graph (or given as `frame_bounds`); edges added later go to the
owner of their source frame.  `rm_keyframe` (frontend only: it renumbers frames, i.e. moves ownership) is not offered.
"""
import numpy as np
import torch
import torch.distributed as dist

from ._cache import tensor_cache_key
from .dist_ba import DistBA, shard_edges_by_source_frame
from .factor_graph import FactorGraph


class DistFactorGraph(FactorGraph):
    def __init__(self, video, update_op, *args, group=None, world=None, rank=None, solver=None, frame_bounds=None, **kw):
        """world / rank default to the process group's; `solver` a DistBA (made here otherwise); `frame_bounds` [world + 1]
        fixes the ownership ranges up front.  Everything else as FactorGraph."""
        super().__init__(video, update_op, *args, **kw)
        self.group = group
        self.world = int(world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1))
        self.rank = int(rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0))
        self.solver = solver if solver is not None else DistBA(self.world, group=group)
        self.bounds = None
        self.frame_lo, self.frame_hi = 0, 1 << 30
        lt = dict(dtype=torch.long, device=self.device)
        self._lii = torch.zeros(0, **lt); self._ljj = torch.zeros(0, **lt)            # this rank's active edges (global order)
        self._lii_inac = torch.zeros(0, **lt); self._ljj_inac = torch.zeros(0, **lt)
        self._had_edges = False
        self._pattern_key = None
        self._kx_key = None
        if frame_bounds is not None:
            self._set_bounds(frame_bounds)

    # ---- ownership ------------------------------------------------------------------------------------------------------
    def _set_bounds(self, bounds):
        bounds = [int(b) for b in bounds]
        assert len(bounds) == self.world + 1 and all(a <= b for a, b in zip(bounds[:-1], bounds[1:])), "frame_bounds: [world + 1], ascending"
        bounds[0], bounds[-1] = 0, 1 << 30
        self.bounds = bounds
        self.frame_lo, self.frame_hi = bounds[self.rank], bounds[self.rank + 1]
        self.solver.set_owned_frames(self.frame_lo, self.frame_hi)                   # collective: all ranks get here together

    def _owned(self, ii):
        return (ii >= self.frame_lo) & (ii < self.frame_hi)

    def _local_edges(self):
        return self._lii, self._ljj

    def _local_inactive(self):
        return self._lii_inac, self._ljj_inac

    def local_index(self):
        """positions of this rank's edges in the global active list"""
        return torch.nonzero(self._owned(self.ii))[:, 0]

    # ---- edge bookkeeping: lists replicated, state local -------------------------------------------------------------
    def add_factors(self, ii, jj, remove=False):
        as_t = lambda x: x.to(self.device, torch.long) if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.long, device=self.device)
        ii, jj = self._filter_repeated_edges(as_t(ii).reshape(-1), as_t(jj).reshape(-1))
        if ii.shape[0] == 0:
            return
        # (FactorGraph tests `self.corr is not None` = "a volume graph that has held edges": the local volume object can be absent
        # on a rank without edges, the replicated flag cannot differ between ranks)
        if self.max_factors > 0 and self.ii.shape[0] + ii.shape[0] > self.max_factors and self._had_edges and self.corr_impl == "volume" and remove:
            ix = torch.argsort(self.age, stable=True)
            self.rm_factors(ix >= self.max_factors - ii.shape[0], store=True)
        if self.bounds is None:
            # ownership ranges from the first edge list: contiguous source-frame ranges balanced by edge count
            _, bounds = shard_edges_by_source_frame(ii.cpu().numpy(), self.world)
            self._set_bounds(bounds)
        own = self._owned(ii)
        li, lj = ii[own], jj[own]
        if li.shape[0] > 0:
            self._append_factors(li, lj)
        self._lii = torch.cat([self._lii, li]); self._ljj = torch.cat([self._ljj, lj])
        self.ii = torch.cat([self.ii, ii]); self.jj = torch.cat([self.jj, jj]); self.age = torch.cat([self.age, torch.zeros_like(ii)])
        self._had_edges = True

    def rm_factors(self, mask, store=False):
        """mask over the GLOBAL edge list (what `graph.age > k`, `graph.ii < k` produce)"""
        if mask.dtype != torch.bool:
            m = torch.zeros(len(self.ii), dtype=torch.bool, device=self.device); m[mask] = True
            mask = m
        own = self._owned(self.ii)
        lmask = mask[own]                                             # the same mask over this rank's edges
        if store:
            self.ii_inac = torch.cat([self.ii_inac, self.ii[mask]]); self.jj_inac = torch.cat([self.jj_inac, self.jj[mask]])
            self._lii_inac = torch.cat([self._lii_inac, self._lii[lmask]]); self._ljj_inac = torch.cat([self._ljj_inac, self._ljj[lmask]])
        if self._lii.shape[0] > 0:
            self._drop_state(lmask, store)
        self._lii = self._lii[~lmask]; self._ljj = self._ljj[~lmask]
        self.ii = self.ii[~mask]; self.jj = self.jj[~mask]; self.age = self.age[~mask]

    def clear_edges(self):
        self.rm_factors(self.ii >= 0)
        self._net = None
        self.corr = None

    def filter_edges(self):
        """factor_graph.py:66-76: the confidence of an edge is a mean over ITS weights -- local to the owner; one all-reduce of
        E floats makes the decision global"""
        conf = torch.zeros(len(self.ii), device=self.device)
        if self._lii.shape[0] > 0:
            conf[self.local_index()] = torch.mean(self.weight, dim=[0, 2, 3, 4])
        self._allreduce(conf)
        mask = (torch.abs(self.ii - self.jj) > 2) & (conf < 0.001)
        self.ii_bad = torch.cat([self.ii_bad, self.ii[mask]])
        self.jj_bad = torch.cat([self.jj_bad, self.jj[mask]])
        self.rm_factors(mask, store=False)

    def rm_keyframe(self, ix):
        raise NotImplementedError("DistFactorGraph.rm_keyframe: removing a keyframe renumbers the frames and with them the ownership "
                                  "of per-edge state; the local-BA frontend that needs it runs on one GPU (FactorGraph)")

    # ---- the solve: this rank's edges into the edge-sharded BA ---------------------------------------------------------
    def _allreduce(self, t):
        if self.world > 1:
            dist.all_reduce(t, group=self.group)

    def _replicated(self, t):
        """a tensor every rank computed from replicated inputs and DECIDES on (the proximity distances): rank 0's copy, so a
        last-bit difference between devices can never make two ranks choose different edges"""
        if self.world > 1:
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        return t

    def _depth_blocks(self, ii, t0, t1):
        """sorted frames whose depth maps this rank's BA call carries a block for: unique(arange(t0, t1) U ii) (ba_cuda,
        src/droid_kernels.cu:1354-1360) -- the rows of `eta`.  Cached per edge-list tensor."""
        key = (tensor_cache_key(ii), int(t0), int(t1))
        if key[0] is None or self._kx_key != key:
            kx = torch.unique(torch.cat([torch.arange(t0, t1, device=self.device), ii]))
            self._kx_key, self._kx = key, (kx, ii)
        return self._kx[0]

    def _set_pattern(self, use_inactive_lists, t0, t1):
        """the co-visible 6x6 blocks of the GLOBAL edge list (replicated, so every rank derives the same pattern) -> the
        all-reduce moves those instead of the dense system; recomputed when the lists or the window change"""
        lists = (self.ii, self.jj) + ((self.ii_inac, self.jj_inac) if use_inactive_lists else ())
        key = (tensor_cache_key(*lists), int(t0), int(t1))
        if key[0] is not None and key == self._pattern_key:
            return
        ii = torch.cat(lists[0::2]).cpu().numpy(); jj = torch.cat(lists[1::2]).cpu().numpy()
        self.solver.set_graph(ii, jj, t0, t1)
        self._pattern_key, self._pattern_lists = key, lists

    def _solve(self, tb, wb, ii, jj, t0, t1, itrs, lm, ep, motion_only, EP, uniq=None):
        v = self.video
        with v.get_lock():
            if t1 is None:
                t1 = self._ba_t1(t0, True)
            # (the pattern over active + inactive edges covers every call: a superset only adds zero blocks to the exchange)
            self._set_pattern(len(self.ii_inac) > 0, t0, t1)
            if motion_only:
                eta = torch.zeros(0, self.ht, self.wd, device=self.device)
            else:
                eta = (.2 * self.damping[self._depth_blocks(ii, t0, t1)] + EP).contiguous()
            self.solver.ba(v.poses, v.disps, v.intrinsics[0].contiguous(), v.disps_sens, tb, wb, eta, ii, jj, t0, t1,
                           itrs, lm, ep, motion_only, alpha=getattr(v, "disps_conf", None))
            v.disps.clamp_(min=0.001)

    # ---- full-resolution depths: computed by the owner, exchanged once per call --------------------------------------
    def _upsample(self, frames, upmask):
        self.video.upsample(frames, upmask)
        self._upsampled = frames if getattr(self, "_upsampled", None) is None else torch.unique(torch.cat([self._upsampled, frames]))

    def _exchange_upsampled(self):
        """every rank upsampled the depth maps of source frames it owns; leave all ranks with all of them (one all-reduce over the
        rows any rank touched: the set is a function of the replicated edge list)"""
        if not self.upsample or self.world == 1:
            self._upsampled = None
            return
        rows = torch.unique(self.ii)
        buf = torch.zeros((len(rows),) + tuple(self.video.disps_up.shape[1:]), device=self.device)
        mine = getattr(self, "_upsampled", None)
        if mine is not None and len(mine) > 0:
            pos = torch.searchsorted(rows, mine)
            buf[pos] = self.video.disps_up[mine]
        touched = torch.zeros(len(rows), device=self.device)
        if mine is not None and len(mine) > 0:
            touched[pos] = 1.0
        self._allreduce(buf); self._allreduce(touched)
        sel = touched > 0
        self.video.disps_up[rows[sel]] = buf[sel]
        self._upsampled = None

    def update(self, *a, **kw):
        if self._lii.shape[0] == 0:
            raise RuntimeError("DistFactorGraph.update: rank %d holds no edges (fewer source frames with edges than ranks)" % self.rank)
        super().update(*a, **kw)
        self._exchange_upsampled()

    def update_lowmem(self, *a, **kw):
        if self._lii.shape[0] == 0:
            raise RuntimeError("DistFactorGraph.update_lowmem: rank %d holds no edges (fewer source frames with edges than ranks)" % self.rank)
        super().update_lowmem(*a, **kw)
        self._exchange_upsampled()
