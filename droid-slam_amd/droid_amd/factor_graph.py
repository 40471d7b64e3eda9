"""FactorGraph on MI355X: edge bookkeeping + the update iteration that drives the hot path.

Host-side mirror of the reference class (droid_slam/factor_graph.py:19-412) with the same constructor, attributes
(`ii, jj, age, corr, net, target [1,E,h,w,2], weight, damping, ii_inac/jj_inac/target_inac/weight_inac, ii_bad/jj_bad`)
and methods (`add_factors, rm_factors, rm_keyframe, filter_edges, clear_edges, update, update_lowmem,
add_neighborhood_factors, add_proximity_factors`).  MI355X-first underneath:

  * one update iteration = reproject (1 fused kernel) -> motion features (1 kernel) -> 4-level pyramid lookup (1 kernel,
    channel-last) -> ConvGRU update operator (droid_amd.update.UpdateModule.forward_nhwc: hidden state kept channel-last
    fp16, context features convolved once per source frame) -> BA inputs (1 kernel) -> droid_backends.ba (all on device);
  * `update_lowmem` runs the on-the-fly correlation on the fp16 MFMA (droid_amd.corr.AltCorrBlock) in source-frame chunks
    (`chunk_frames`, default 8 like the reference; with 288 GB of HBM much larger chunks fit) and ONE BA over all edges;
  * `add_proximity_factors` keeps the t x t distance matrix on the device: masking, suppression around existing edges and
    the greedy NMS walk are kernels (droid_backends.proximity_nms), one 4-byte read-back tells how many edges were chosen.
The reference copies the matrix to the CPU and loops over it in Python (factor_graph.py:363-409).
"""
import os
import time

import torch

import droid_backends as db
from .corr import CorrBlock, CorrBlockRef, AltCorrBlock
from ._cache import tensor_cache_key
from .trace import roctx_range


class FactorGraph:
    def __init__(self, video, update_op, device="cuda", corr_impl="volume", max_factors=-1, upsample=False, chunk_frames=8, native_corr=None,
                 coherence_fallback=True):
        self.video = video
        self.update_op = update_op
        self.device = video.device if hasattr(video, "device") else torch.device(device)
        self.max_factors = max_factors
        self.corr_impl = corr_impl
        self.upsample = upsample
        self.chunk_frames = chunk_frames
        self.ht = ht = video.ht // 8
        self.wd = wd = video.wd // 8
        lt = dict(dtype=torch.long, device=self.device)
        self.ii = torch.zeros(0, **lt); self.jj = torch.zeros(0, **lt); self.age = torch.zeros(0, **lt)
        self.corr = None
        self._net = None                                              # [E,h,w,128] fp16, channel-last
        self._glo = None                                              # (the _net object, its global-context sums [E,128] f32): see _operator
        # GraphAgg's upmask head is only read by DepthVideo.upsample: None = compute it iff this graph upsamples (the reference computes it
        # in every iteration and drops it when upsample=False, its callers' default); True / False force it (bench.py forces True)
        self.compute_upmask = None
        self.damping = 1e-6 * torch.ones_like(self.video.disps)
        self.target = torch.zeros(1, 0, ht, wd, 2, device=self.device)
        self.weight = torch.zeros(1, 0, ht, wd, 2, device=self.device)
        self.ii_inac = torch.zeros(0, **lt); self.jj_inac = torch.zeros(0, **lt)
        self.ii_bad = torch.zeros(0, **lt); self.jj_bad = torch.zeros(0, **lt)
        self.target_inac = torch.zeros(1, 0, ht, wd, 2, device=self.device)
        self.weight_inac = torch.zeros(1, 0, ht, wd, 2, device=self.device)
        # the MI355X pyramid (any image up to 64 columns wide: sizes outside its layout sit on a zero-padded canvas, see
        # droid_amd.corr.CorrBlock); native_corr=False forces the reference-layout volumes (CorrBlockRef), the only form for wider images
        self._native_corr = CorrBlock.supported(ht, wd) if native_corr is None else (bool(native_corr) and CorrBlock.supported(ht, wd))
        # The pyramid layout coalesces when the 64 pixels of an 8x8 block look at the same displacement cells (every flow a
        # reprojection produces); for an INCOHERENT flow it is slower than the reference layout.  update() therefore measures the
        # spread of the flow once per edge list (CorrBlock.window_spread, one read-back) and, beyond CorrBlock.SPREAD_LIMIT, rebuilds
        # the volumes in the reference layout and stays there.  native_corr=True pins the pyramid; coherence_fallback=False skips the check.
        self._coherence_fallback = bool(coherence_fallback) and native_corr is None
        self._coherence_key = None
        # the gates' per-frame context term is kept between keyframe insertions (_context); False recomputes it in every update
        # iteration, like the reference's convolutions over [net, inp, corr, flow] do (bench.py times that form)
        self.cache_context = True
        self.phase_events = None        # a list: update() appends (name, start, end) torch.cuda.Event triples (bench.py)
        self._stats_key = None

    # ---- which edges' state this object holds ---------------------------------------------------------------------
    # `ii, jj, age, ii_inac, jj_inac` are the graph's edge lists.  The per-edge state (hidden state, targets, weights, volumes)
    # belongs to the edges `_local_edges()` returns: all of them here, this rank's shard in droid_amd.dist_graph.DistFactorGraph.
    def _local_edges(self):
        return self.ii, self.jj

    def _local_inactive(self):
        return self.ii_inac, self.jj_inac

    def _edge_stats(self):
        """(min source frame, max frame + 1) of the edge list, read back ONCE per edge list (the reference does `.item()` /
        python `range(tensor)` on them in every update, factor_graph.py:228,284)"""
        key = tensor_cache_key(self.ii, self.jj)
        if key is None or key != self._stats_key:
            r = torch.stack([self.ii.min(), self.ii.max(), self.jj.max()]).tolist()
            self._stats_key, self._stats = key, (int(r[0]), int(r[1]), int(r[2]), self.ii, self.jj)
        return self._stats[:3]

    def _replicated(self, t):
        return t                                    # (one process: droid_amd.dist_graph broadcasts rank 0's copy)

    def _unique_cached(self, ii):
        """torch.unique(ii) (a read-back: the output size is data dependent), once per edge-list tensor"""
        key = tensor_cache_key(ii)
        if key is None or getattr(self, "_uniq_key", None) != key:
            self._uniq_key, self._uniq = key, (torch.unique(ii), ii)
        return self._uniq[0]

    def _solve(self, tb, wb, ii, jj, t0, t1, itrs, lm, ep, motion_only, EP, uniq=None):
        """eta of the depth blocks (factor_graph.py:251) + the dense BA over the edges (ii, jj) with targets / weights (tb, wb)"""
        if uniq is None:
            uniq = self._unique_cached(ii)
        eta = (.2 * self.damping[uniq] + EP).contiguous()
        self.video.ba(tb, wb, eta, ii, jj, t0, t1, itrs=itrs, lm=lm, ep=ep, motion_only=motion_only)

    # ---- reference-shaped views of the channel-last state ---------------------------------------------------------
    @property
    def net(self):
        self._glo = None        # (a caller holding this view may write through it: the sums kept for the next iteration are dropped)
        return None if self._net is None else self._net.permute(0, 3, 1, 2)[None]        # [1,E,128,h,w]

    @property
    def inp(self):
        return self.video.inps[self.ii][None]                                                # factor_graph.py:135

    # ---- edge bookkeeping (factor_graph.py:52-212) ------------------------------------------------------------------
    def _filter_repeated_edges(self, ii, jj):
        for a, b in ((self.ii, self.jj), (self.ii_inac, self.jj_inac)):
            if len(a) > 0 and len(ii) > 0:
                m = ((ii[:, None] == a) & (jj[:, None] == b)).any(dim=-1)
                ii, jj = ii[~m], jj[~m]
        return ii, jj

    def filter_edges(self):
        conf = torch.mean(self.weight, dim=[0, 2, 3, 4])
        mask = (torch.abs(self.ii - self.jj) > 2) & (conf < 0.001)
        self.ii_bad = torch.cat([self.ii_bad, self.ii[mask]])
        self.jj_bad = torch.cat([self.jj_bad, self.jj[mask]])
        self.rm_factors(mask, store=False)

    def clear_edges(self):
        self.rm_factors(self.ii >= 0)
        self._net = None
        self._glo = None

    def add_factors(self, ii, jj, remove=False):
        as_t = lambda x: x.to(self.device, torch.long) if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.long, device=self.device)
        ii, jj = self._filter_repeated_edges(as_t(ii).reshape(-1), as_t(jj).reshape(-1))
        if ii.shape[0] == 0:
            return
        if self.max_factors > 0 and self.ii.shape[0] + ii.shape[0] > self.max_factors and self.corr is not None and remove:
            ix = torch.argsort(self.age, stable=True)                 # factor_graph.py:121-122 (mask indexed by edge position)
            self.rm_factors(ix >= self.max_factors - ii.shape[0], store=True)
        self._append_factors(ii, jj)
        self.ii = torch.cat([self.ii, ii]); self.jj = torch.cat([self.jj, jj]); self.age = torch.cat([self.age, torch.zeros_like(ii)])

    def _append_factors(self, ii, jj):
        """per-edge state of new edges: hidden state from the source frame's context network output, correlation volumes, the
        first target = the reprojection (factor_graph.py:124-145)"""
        net = self.video.nets[ii].permute(0, 2, 3, 1).contiguous()
        if self.corr_impl == "volume":
            if self._native_corr:                                     # features prepared once per frame, indexed by the edges
                corr = CorrBlock.from_frames(self.video.fmaps, ii, jj)
            else:
                c = (ii == jj).long()
                corr = CorrBlockRef(self.video.fmaps[ii, 0][None], self.video.fmaps[jj, c][None])
            self.corr = corr if self.corr is None else self.corr.cat(corr)
        target, _ = self.video.reproject(ii, jj)
        self._net = net if self._net is None else torch.cat([self._net, net], 0)
        self._glo = None                                              # (sums of the replaced state tensor: release it)
        self.target = torch.cat([self.target, target], 1)
        self.weight = torch.cat([self.weight, torch.zeros_like(target)], 1)

    def rm_factors(self, mask, store=False):
        if store:
            self.ii_inac = torch.cat([self.ii_inac, self.ii[mask]]); self.jj_inac = torch.cat([self.jj_inac, self.jj[mask]])
        self._drop_state(mask, store)
        self.ii = self.ii[~mask]; self.jj = self.jj[~mask]; self.age = self.age[~mask]

    def _drop_state(self, mask, store):
        """per-edge state of removed edges (mask over the edges whose state this object holds)"""
        if store:
            self.target_inac = torch.cat([self.target_inac, self.target[:, mask]], 1)
            self.weight_inac = torch.cat([self.weight_inac, self.weight[:, mask]], 1)
        if self.corr_impl == "volume" and self.corr is not None:
            self.corr = self.corr[~mask]
        if self._net is not None:
            self._net = self._net[~mask]
            self._glo = None
        self.target = self.target[:, ~mask]; self.weight = self.weight[:, ~mask]

    def rm_keyframe(self, ix):
        """drop keyframe ix: shift the video buffers down and re-index the edges (factor_graph.py:182-212)"""
        v = self.video
        t = v.counter.value
        for name in ("images", "poses", "disps", "disps_sens", "intrinsics", "nets", "inps", "fmaps", "tstamp"):
            buf = getattr(v, name)
            buf[ix:t - 1] = buf[ix + 1:t].clone()
        m = (self.ii_inac == ix) | (self.jj_inac == ix)
        self.ii_inac[self.ii_inac >= ix] -= 1; self.jj_inac[self.jj_inac >= ix] -= 1
        if torch.any(m):
            self.ii_inac = self.ii_inac[~m]; self.jj_inac = self.jj_inac[~m]
            self.target_inac = self.target_inac[:, ~m]; self.weight_inac = self.weight_inac[:, ~m]
        m = (self.ii == ix) | (self.jj == ix)
        self.ii[self.ii >= ix] -= 1; self.jj[self.jj >= ix] -= 1
        self.rm_factors(m, store=False)

    # ---- the update iteration (factor_graph.py:214-263) ------------------------------------------------------------
    def _operator(self, net, coords1, target_prev, feats, ii, corr0=None):
        """motion features + update operator on a set of edges -> (dw [E,h,w,4], damping [K,h,w], upmask, uniq)"""
        flow = db.motion_features(coords1, target_prev)
        uniq, ix, inp_frames, ctx = self._context(ii)
        # Round 6: the ConvGRU's global-context reduction of the state this call writes is computed inside its q-gate launch and handed
        # to the NEXT call on the same tensor (gru.py:23-24 reads what gru.py:31 wrote).  The pair (tensor object, sums) is only reused
        # while `net` IS that object: every edit of the edge set replaces self._net (cat / mask indexing in _append_factors / _drop_state),
        # and nothing but the operator writes into it in place; chunk views (update_lowmem) are new objects and never match.
        chain = net is self._net and hasattr(self.update_op, "fuses_next_glo")        # (an operator object without the extension: plain call)
        glo = self._glo[1] if (chain and self._glo is not None and self._glo[0] is net) else None
        kw = dict(glo_red=glo, glo_next=True) if chain else {}
        if hasattr(self.update_op, "fuses_next_glo"):
            kw["want_upmask"] = bool(self.upsample if self.compute_upmask is None else self.compute_upmask)
        _, _, _, damping, upmask = self.update_op.forward_nhwc(net, None, feats, flow, ii, inp_frames=inp_frames, inp_index=ix, ctx=ctx,
                                                               corr0=corr0, **kw)
        self._glo = (net, self.update_op.last_glo) if (chain and self.update_op.last_glo is not None) else None
        return self.update_op.last_dw, damping, upmask, uniq

    def _pyramid_features(self, block, coords1):
        """correlation features of all edges from a CorrBlock -> (feats, corr0) for _operator: by default the lookup runs fused
        with the correlation encoder's first layer (CorrBlock.lookup_corr0; option lookup_fused), else as the reference-layout
        / channel-last lookup that the operator's first layer reads back"""
        if db.get_option("lookup_fused"):
            return None, block.lookup_corr0(coords1[None], self.update_op)
        if self.update_op.wants_reference_layout_corr(*coords1.shape[1:3]):
            return block(coords1[None])[0], None
        return block.lookup_nhwc(coords1[None]), None

    def _context(self, ii):
        """source frames of the edge list, their context features channel-last and the gates' per-frame context term
        (UpdateModule.context_term).  All three depend only on the edge list and on video.inps, which change when a keyframe is
        added or removed, not between the update iterations in between: kept until either changes (tensor version counters)."""
        inps = self.video.inps
        key = tensor_cache_key(ii, inps)
        if key is None or not self.cache_context or getattr(self, "_ctx_key", None) != key:
            uniq, ix = torch.unique(ii, return_inverse=True)
            inp_frames = inps[uniq].permute(0, 2, 3, 1).contiguous()
            h, w = inp_frames.shape[1:3]
            ctx = self.update_op.context_term(inp_frames) if (w == 64 and h % 4 == 0) else None
            self._ctx_key, self._ctx = key, (uniq, ix.contiguous(), inp_frames, ctx, ii)     # (ii kept alive: its address is in the key)
        return self._ctx[:4]

    def _check_flow_coherence(self, coords1):
        """once per edge list: if the flow's window spread says the pyramid layout would be slower than the reference layout,
        rebuild every edge's volumes as CorrBlockRef (sticky for this graph; the pyramid is released first)"""
        ii, jj = self._local_edges()
        key = tensor_cache_key(ii, jj)
        if key is not None and key == self._coherence_key:
            return
        self._coherence_key, self._coherence_ii = key, (ii, jj)                    # (kept alive: their addresses are in the key)
        # (a mean over at most 256 evenly spaced edges: the figure decides between two layouts a factor 6 apart, and the frontend
        # changes its edge list -- and so comes through here -- at every keyframe)
        self.last_window_spread = spread = CorrBlock.window_spread(coords1[::max(1, coords1.shape[0] // 256)])
        if spread <= CorrBlock.SPREAD_LIMIT:
            return
        c = (ii == jj).long()
        self.corr = None                                                            # 105 GB at C3: release before the rebuild
        torch.cuda.empty_cache()
        self.corr = CorrBlockRef(self.video.fmaps[ii, 0][None], self.video.fmaps[jj, c][None])
        self._native_corr = False

    def _ba_t1(self, t0, use_inactive):
        """the BA window's end when the caller gives none: one past the largest frame index of the BA's edges
        (depth_video.py:218-219), from the cached edge statistics instead of two read-backs per call"""
        _, imax, jmax = self._edge_stats()
        t1 = max(imax, jmax) + 1
        if use_inactive and len(self.ii_inac) > 0:
            key = (tensor_cache_key(self.ii_inac, self.jj_inac), t0)
            if key[0] is None or getattr(self, "_inac_t1_key", None) != key:
                m = (self.ii_inac >= t0 - 3) & (self.jj_inac >= t0 - 3)
                top = torch.where(m, torch.maximum(self.ii_inac, self.jj_inac), torch.full_like(self.ii_inac, -1)).max()
                self._inac_t1_key, self._inac_t1 = key, (int(top.item()) + 1, self.ii_inac, self.jj_inac)
            t1 = max(t1, self._inac_t1[0])
        return t1

    def update(self, t0=None, t1=None, itrs=2, use_inactive=False, EP=1e-7, motion_only=False, lm=1e-4, ep=0.1):
        """one update iteration (factor_graph.py:214-263).  `lm`, `ep`: the BA's damping (the reference's update() fixes 1e-4 / 0.1,
        its update_lowmem 1e-5 / 1e-2; bench.py runs the global-BA parameters through this full-batch form)."""
        ii, jj = self._local_edges()
        ev = self.phase_events

        def mark():
            if ev is None:
                return None
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        with roctx_range("droid.update/reproject"):
            coords1 = self.video.reproject(ii, jj)[0][0]                                      # [E,h,w,2]
        if self.corr_impl != "volume":
            raise RuntimeError("update() needs corr_impl='volume' (use update_lowmem for 'alt')")
        e0 = mark()
        with roctx_range("droid.update/corr_lookup"):
            if self._native_corr and self._coherence_fallback:
                self._check_flow_coherence(coords1)
            if self._native_corr:
                feats, corr0 = self._pyramid_features(self.corr, coords1)
            else:
                feats, corr0 = self.update_op.corr_to_nhwc(self.corr(coords1[None])[0]), None
        e1 = mark()
        with roctx_range("droid.update/update_operator"):
            dw, damping, upmask, uniq = self._operator(self._net, coords1, self.target[0].contiguous(), feats, ii, corr0)
        if t0 is None:
            t0 = max(1, self._edge_stats()[0] + 1)
        if t1 is None:
            t1 = self._ba_t1(t0, use_inactive)
        with roctx_range("droid.update/ba"):
            target, weight, tb, wb = db.ba_inputs(coords1, dw)
            self.target, self.weight = target[None], weight[None]
            self.damping[uniq] = damping
            uniq_ba = uniq
            if use_inactive:
                ii_in, jj_in = self._local_inactive()
                m = (ii_in >= t0 - 3) & (jj_in >= t0 - 3)
                ii = torch.cat([ii_in[m], ii]); jj = torch.cat([jj_in[m], jj])
                tb = torch.cat([self.target_inac[0, m].permute(0, 3, 1, 2), tb]).contiguous()
                wb = torch.cat([self.weight_inac[0, m].permute(0, 3, 1, 2), wb]).contiguous()
                uniq_ba = None
            e2 = mark()
            self._solve(tb, wb, ii, jj, t0, t1, itrs, lm, ep, motion_only, EP, uniq=uniq_ba)
        e3 = mark()
        if ev is not None:
            ev.append((e0, e1, e2, e3))
        if self.upsample:
            with roctx_range("droid.update/upsample"):
                self._upsample(uniq, upmask)
        self.age += 1

    def _upsample(self, frames, upmask):
        self.video.upsample(frames, upmask)

    # ---- global BA (factor_graph.py:266-330) --------------------------------------------------------------------------
    def _pyramid_fits(self, E, ht, wd):
        """Deterministic rule of update_lowmem(corr="auto"): a function of the problem size and the DEVICE (its total HBM), not
        of what happens to be free at call time -- two runs of the same sequence take the same path and round the same way.
        The materialised pyramid of all E edges plus the full-batch activations of the update operator (~3.5 KB per
        edge-pixel) must fit into 60 % of the device memory (288 GB per MI355X: 512 keyframes / 4096 edges at 48x64 need
        105 + 44 GB).  If the build then runs out of memory anyway (other tenants of the device), update_lowmem falls back to
        the alt-correlation loop."""
        if not CorrBlock.supported(ht, wd):
            return False
        total = torch.cuda.get_device_properties(self.device).total_memory
        need = E * CorrBlock.bytes_per_edge(ht, wd) + E * ht * wd * 3584
        return need < 0.6 * total

    def _check_activations_fit(self, E, ht, wd):
        need = E * ht * wd * 3584
        free_b = torch.cuda.mem_get_info(self.device)[0] + torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
        if free_b < need:
            raise torch.cuda.OutOfMemoryError("update_lowmem: %.1f GB of operator activations do not fit next to the pyramid (%.1f GB free)" % (need / 1e9, free_b / 1e9))

    def _pyramid_arena(self, E, ht, wd):
        """storage for update_lowmem's per-call pyramid, kept across calls (reallocated only when a call has more edges than any
        before it; released by release_pyramid_arena()): the pyramid is rebuilt on every call because the graph changes between calls, its memory
        need not be -- a cold 105 GB allocation takes seconds (bench.py `ms_pyramid_alloc`), the build 41 ms"""
        a = getattr(self, "_arena", None)
        rec = CorrBlock.bytes_per_edge(ht, wd) // 2
        if a is None or a.shape[0] < E or a.shape[1] != rec:
            self._arena = a = None
            self._arena = a = CorrBlock.arena(E, ht, wd, self.device)
        return a

    def release_pyramid_arena(self):
        self._arena = None

    def _ba_global(self, tb, wb, itrs, use_inactive, EP, t):
        ii, jj = self._local_edges()
        if use_inactive:
            ii_in, jj_in = self._local_inactive()
            ii = torch.cat([ii_in, ii]); jj = torch.cat([jj_in, jj])
            tb = torch.cat([self.target_inac[0].permute(0, 3, 1, 2), tb]).contiguous()
            wb = torch.cat([self.weight_inac[0].permute(0, 3, 1, 2), wb]).contiguous()
        self.age += 1
        with roctx_range("droid.update_lowmem/ba"):
            self._solve(tb, wb, ii, jj, 1, t, itrs, 1e-5, 1e-2, False, EP)
        self.video.dirty[:t] = True

    def update_lowmem(self, t0=None, t1=None, itrs=2, use_inactive=False, EP=1e-7, steps=8, corr="auto"):
        """`steps` global-BA iterations over all edges (factor_graph.py:266-330).  Correlation features, `corr` =
          "alt"      the reference's scheme: on-the-fly correlation (MFMA alt-corr kernel) in chunks of `chunk_frames` source
                     frames, nothing materialised;
          "pyramid"  the 4-level pyramid of every edge is built ONCE for the whole call (fmaps do not change during it) and
                     each step is one full-batch lookup + update operator, like update(): the build is amortised over the
                     steps and the lookup is HBM-bound instead of recomputing 8x8 x 128-channel dot products per step;
          "auto"     "pyramid" when pyramid + activations fit into 60 % of the DEVICE's memory (a rule of E, ht, wd and the
                     device only: deterministic across runs; 105 + 44 GB for 4096 edges at 48x64), else "alt"; an
                     out-of-memory error during the build also falls back to "alt".
        Both paths skip the edges the reference's chunk loop never visits (source frame beyond the last chunk).
        The pyramid's CONTENT is dropped when the call returns; its storage is kept for the next call (_pyramid_arena)."""
        v = self.video
        t = v.counter.value
        num, rig, ch, ht, wd = v.fmaps.shape
        s = self.chunk_frames
        lii, ljj = self._local_edges()
        # the reference's chunk loop (factor_graph.py:284-287) runs over source frames [ii.min(), jj.max()] in steps of s: edges
        # whose source frame lies beyond the last chunk are never visited and keep their previous hidden state / target /
        # weight.  Both correlation paths reproduce that, so they are equivalent up to fp16 rounding of the features.
        imin, imax, jmax = self._edge_stats()
        lo, hi = imin, jmax + 1
        limit = lo + (hi - lo + s - 1) // s * s
        all_visited = imax < limit
        if len(lii) > 0 and (corr == "pyramid" or (corr == "auto" and self._pyramid_fits(len(lii), ht, wd))):
            sel = None if all_visited else torch.nonzero(lii < limit)[:, 0]
            ii_v, jj_v = (lii, ljj) if sel is None else (lii[sel], ljj[sel])
            trace = os.environ.get("DH_LOWMEM_TRACE", "0") == "1"      # wall-clock phases of a call (diagnostics; synchronises)
            if trace:
                torch.cuda.synchronize(); _t0 = time.perf_counter()
            arena = None
            try:
                with roctx_range("droid.update_lowmem/pyramid_build"):
                    arena = self._pyramid_arena(len(ii_v), ht, wd) if CorrBlock.strip_bounds(ht, wd) is None else None
                    block = CorrBlock.from_frames(v.fmaps, ii_v, jj_v, out=arena)
                # the update operator's full-batch activations (~3.5 KB per edge-pixel) are allocated inside the first step: check
                # now, while nothing of this call has been written, that they fit -- what the driver reports free plus what torch's
                # allocator holds unused -- so that a device shared with other tenants still falls back to the alt-correlation loop
                # with the state untouched.  (No trial allocation: a 45 GB probe block fragments the allocator's segments and cost
                # the NEXT call a fresh 1.2 s hipMalloc -- measured, round 5.)
                self._check_activations_fit(len(ii_v), ht, wd)
            except torch.cuda.OutOfMemoryError:
                if corr == "pyramid":
                    raise
                # "auto": fall through to the alt-correlation loop -- with the pyramid's storage really released (the local
                # `arena` is the last reference to ~105 GB at C3: without dropping it empty_cache() frees nothing)
                block = None
                arena = None
                self._arena = None
                torch.cuda.empty_cache()
            del arena
            if trace:
                torch.cuda.synchronize(); _t1 = time.perf_counter()
                print("update_lowmem trace: pyramid build + reservation %.1f ms (block %s)" % (1e3 * (_t1 - _t0), "ok" if block is not None else "FELL BACK"), flush=True)
            if block is not None:
                for _ in range(steps):
                    if trace:
                        torch.cuda.synchronize(); _t2 = time.perf_counter()
                    with roctx_range("droid.update_lowmem/reproject"):
                        coords_all = v.reproject(lii, ljj)[0][0]
                    if sel is None:
                        coords1, net, target_prev = coords_all, self._net, self.target[0].contiguous()
                    else:
                        coords1, net, target_prev = coords_all[sel].contiguous(), self._net[sel].contiguous(), self.target[0][sel].contiguous()
                    with roctx_range("droid.update_lowmem/corr_lookup"):
                        feats, corr0 = self._pyramid_features(block, coords1)
                    with roctx_range("droid.update_lowmem/update_operator"):
                        dw, damping, upmask, uniq = self._operator(net, coords1, target_prev, feats, ii_v, corr0)
                    target, weight, tb, wb = db.ba_inputs(coords1, dw)
                    if sel is not None:                                 # unvisited edges keep what they had
                        self._net[sel] = net
                        t_all, w_all = self.target[0].clone(), self.weight[0].clone()
                        t_all[sel] = target; w_all[sel] = weight
                        target, weight = t_all, w_all
                        tb = target.permute(0, 3, 1, 2).contiguous(); wb = weight.permute(0, 3, 1, 2).contiguous()
                    self.target, self.weight = target[None], weight[None]
                    self.damping[uniq] = damping
                    if self.upsample:
                        self._upsample(uniq, upmask)
                    self._ba_global(tb, wb, itrs, use_inactive, EP, t)
                    if trace:
                        torch.cuda.synchronize(); print("update_lowmem trace: step %.1f ms" % (1e3 * (time.perf_counter() - _t2)), flush=True)
                return
        corr_op = AltCorrBlock(v.fmaps.view(1, num * rig, ch, ht, wd))
        for _ in range(steps):
            coords1 = v.reproject(lii, ljj)[0][0]
            target_prev = self.target[0].contiguous()
            if not all_visited:
                # edges beyond the last chunk are not visited and keep their previous target / weight (see above)
                target = target_prev.clone(); weight = self.weight[0].clone()
                tb = target.permute(0, 3, 1, 2).contiguous(); wb = weight.permute(0, 3, 1, 2).contiguous()
            else:                                                       # every edge is written below
                target = torch.empty_like(target_prev); weight = torch.empty_like(target_prev)
                tb = torch.empty(len(lii), 2, ht, wd, device=self.device); wb = torch.empty_like(tb)
            for i in range(lo, hi, s):
                vmask = (lii >= i) & (lii < i + s)
                e = torch.nonzero(vmask)[:, 0]
                if e.numel() == 0:
                    continue
                iis, jjs = lii[e], ljj[e]
                c1 = coords1[e].contiguous()
                corr1 = corr_op(c1[None], rig * iis, rig * jjs + (iis == jjs).long())         # [1,M,196,h,w]
                feats = corr1[0] if self.update_op.wants_reference_layout_corr(ht, wd) else self.update_op.corr_to_nhwc(corr1[0])
                net = self._net[e].contiguous()
                dw, damping, upmask, uniq = self._operator(net, c1, target_prev[e].contiguous(), feats, iis)
                if self.upsample:
                    self._upsample(uniq, upmask)
                self._net[e] = net
                tg, wg, tbe, wbe = db.ba_inputs(c1, dw)
                target[e] = tg; weight[e] = wg; tb[e] = tbe; wb[e] = wbe
                self.damping[uniq] = damping
            self.target, self.weight = target[None], weight[None]
            self._ba_global(tb, wb, itrs, use_inactive, EP, t)

    # ---- edge creation policies (factor_graph.py:332-412) ---------------------------------------------------------------
    def add_neighborhood_factors(self, t0, t1, r=3):
        ii, jj = torch.meshgrid(torch.arange(t0, t1, device=self.device), torch.arange(t0, t1, device=self.device), indexing="ij")
        c = 1 if self.video.stereo else 0
        keep = ((ii - jj).abs() > c) & ((ii - jj).abs() <= r)
        self.add_factors(ii[keep], jj[keep])

    def add_proximity_factors(self, t0=0, t1=0, rad=2, nms=2, beta=0.25, thresh=16.0, remove=False):
        v = self.video
        t = v.counter.value
        ix = torch.arange(t0, t, device=self.device); jx = torch.arange(t1, t, device=self.device)
        ii, jj = torch.meshgrid(ix, jx, indexing="ij")
        d = self._replicated(v.distance(ii.reshape(-1), jj.reshape(-1), beta=beta).contiguous())
        # edges that are always added (known without looking at the distances): stereo self edges + temporal neighbours
        es = []
        for i in range(t0, t):
            if v.stereo:
                es.append((i, i))
            for j in range(max(i - rad - 1, 0), i):
                es.append((i, j)); es.append((j, i))
        ei = torch.cat([self.ii, self.ii_bad, self.ii_inac]).contiguous(); ej = torch.cat([self.jj, self.jj_bad, self.jj_inac]).contiguous()
        n_cand = d.numel()
        max_new = max(1, min(n_cand, 1 << 16))
        new_edges, count = db.proximity_nms(d, ei, ej, t0, t1, t, rad, nms, float(thresh), int(self.max_factors), len(es),
                                            bool(v.stereo), max_new)
        n = int(count.item())                                         # the one read-back: how many edges were chosen
        fixed = torch.as_tensor(es, dtype=torch.long, device=self.device).reshape(-1, 2)
        edges = torch.cat([fixed, new_edges[:2 * n]], 0)
        if edges.shape[0] > 0:
            self.add_factors(edges[:, 0], edges[:, 1], remove)
