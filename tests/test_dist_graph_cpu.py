"""world_size-2 gloo test of the HOST logic of the edge-sharded factor graph (droid_amd/dist_graph.py) on CPU.

The kernels need a GPU; here every droid_backends entry point the graph classes call is replaced by a small deterministic torch
function ("recording backend": same signatures, CPU tensors), the update operator by a stand-in whose per-edge outputs depend on
the edge's own inputs and whose per-frame damping is a mean over the edges of the source frame (GraphAgg's structure), and the BA by
the oracle (the two halves DistBA exchanges between: tests/test_dist_cpu.py::OracleBackend).  What runs unchanged is everything this
round added between the kernels: ownership by source frame, replicated edge lists vs rank-local per-edge state across add_factors /
rm_factors(store) / use_inactive / clear_edges, the rows of eta a rank's BA call carries, the block pattern of the exchange, the
depth-confidence map reaching the sharded build, and update / update_lowmem ending with identical poses / depths on all ranks.
Sharded result == the single-process FactorGraph on the same fakes.
"""
import os
import sys
import types
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "droid-slam_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HT, WD, NF = 8, 12, 7


class FakeDB:
    """droid_backends' graph-side entry points on CPU tensors (deterministic stand-ins, not the kernels' arithmetic)"""

    @staticmethod
    def get_option(name):
        return 0                                              # lookup_fused = 0: the un-fused lookup path

    @staticmethod
    def reproject(poses, disps, intrinsics, ii, jj):
        E, (h, w) = len(ii), disps.shape[1:]
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        grid = torch.stack([xx, yy], -1)[None]
        shift = (poses[jj, :2] - poses[ii, :2])[:, None, None, :] * disps[ii][..., None] * 4.0
        return (grid + shift).contiguous(), torch.ones(E, h, w, 1)

    @staticmethod
    def motion_features(coords1, target):
        yy, xx = torch.meshgrid(torch.arange(coords1.shape[1], dtype=torch.float32), torch.arange(coords1.shape[2], dtype=torch.float32), indexing="ij")
        c0 = torch.stack([xx, yy], -1)[None]
        return torch.cat([coords1 - c0, target - coords1], -1).clamp(-64, 64)

    @staticmethod
    def ba_inputs(coords1, dw):
        target = coords1 + dw[..., :2]
        weight = dw[..., 2:]
        return target, weight, target.permute(0, 3, 1, 2).contiguous(), weight.permute(0, 3, 1, 2).contiguous()


class FakeCorr:
    """AltCorrBlock / CorrBlock stand-in: features of an edge = a function of its two frame indices and its coordinates"""

    def __init__(self, *a, **kw):
        pass

    def __call__(self, coords, ii, jj):
        c = coords[0]
        f = torch.sin(c.sum(-1) * 0.1 + (ii * 0.37 + jj * 0.11)[:, None, None].float())
        return f[None, :, None].expand(1, len(ii), 4, *c.shape[1:3]).contiguous()


class FakeCorrRef:
    """CorrBlockRef stand-in (the volume form update() needs): per-edge state that cat / boolean indexing must keep aligned"""

    def __init__(self, fmap1, fmap2):
        self.tag = fmap1[0].float().mean((1, 2, 3)) * 0.37 + fmap2[0].float().mean((1, 2, 3)) * 0.11      # [E]

    def __call__(self, coords):
        c = coords[0]
        f = torch.sin(c.sum(-1) * 0.1 + self.tag[:, None, None])
        return f[None, :, None].expand(1, len(self.tag), 4, *c.shape[1:3]).contiguous()

    def cat(self, other):
        self.tag = torch.cat([self.tag, other.tag])
        return self

    def __getitem__(self, index):
        self.tag = self.tag[index]
        return self


class FakeUpdate:
    """update-operator stand-in with the real one's interface towards FactorGraph._operator"""

    def wants_reference_layout_corr(self, h, w):
        return True

    def corr_to_nhwc(self, x):
        return x

    def context_term(self, inp_frames):
        return None

    def forward_nhwc(self, net, inp, feats, flow, ii, inp_frames=None, inp_index=None, ctx=None, corr0=None):
        f = feats[:, 0]                                            # [E,h,w]
        ctxv = inp_frames[inp_index].float().mean(-1)              # the source frame's context features reach its edges
        net.mul_(0.5).add_((0.25 * torch.tanh(f + ctxv))[..., None].to(net.dtype))        # hidden state, in place
        nf = net.float().mean(-1)
        delta = torch.stack([0.3 * torch.tanh(nf + flow[..., 0] * 0.05), 0.2 * torch.tanh(nf - flow[..., 3] * 0.05)], -1)
        wgt = torch.sigmoid(torch.stack([nf + f, nf - f], -1))
        self.last_dw = torch.cat([delta, wgt], -1).float().contiguous()
        uniq, inv = torch.unique(ii, return_inverse=True)
        s = torch.zeros(len(uniq), *nf.shape[1:]).index_add_(0, inv, torch.sigmoid(nf))
        cnt = torch.zeros(len(uniq)).index_add_(0, inv, torch.ones(len(ii)))
        damping = 1e-3 * s / cnt[:, None, None]                    # GraphAgg: a mean over the edges of a source frame
        return None, None, None, damping, torch.zeros(len(uniq), *nf.shape[1:], 1)


def _install_fakes():
    from droid_amd import factor_graph as fg_mod, depth_video as dv_mod
    fg_mod.db = FakeDB
    dv_mod.db = FakeDB
    fg_mod.AltCorrBlock = FakeCorr
    fg_mod.CorrBlockRef = FakeCorrRef


def _scenario():
    """poses / depths / features of a small graph + the edit script both runs follow"""
    from droid_amd import synthetic as syn
    g = syn.small_graph(n_frames=NF, seed=11, ht=HT, wd=WD, radius=3, sensor_depth=True)
    rng = np.random.default_rng(5)
    g["nets"] = np.tanh(rng.standard_normal((NF, 16, HT, WD))).astype(np.float32)
    g["inps"] = np.maximum(rng.standard_normal((NF, 16, HT, WD)), 0).astype(np.float32)
    g["conf"] = rng.uniform(0.0, 0.2, (NF, HT, WD)).astype(np.float32)
    return g


def _make_video(g, with_conf):
    from droid_amd.depth_video import DepthVideo
    v = DepthVideo(image_size=[8 * HT, 8 * WD], buffer=NF + 1, stereo=False, device="cpu")
    v.nets = torch.zeros(NF + 1, 16, HT, WD); v.inps = torch.zeros(NF + 1, 16, HT, WD)          # (16 channels keep the fakes small)
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a))
    v.poses[:NF] = T(g["poses"]); v.disps[:NF] = T(g["disps"]); v.disps_sens[:NF] = T(g["disps_sens"])
    v.intrinsics[:NF] = T(g["intrinsics"]); v.nets[:NF] = T(g["nets"]); v.inps[:NF] = T(g["inps"])
    v.fmaps[:NF] = torch.arange(NF, dtype=torch.float16)[:, None, None, None, None]      # feature maps that identify their frame
    v.counter.value = NF
    if with_conf:
        v.set_depth_confidence(slice(0, NF), T(g["conf"]))
    return v


def _script(graph, lowmem):
    """the same sequence of graph edits and iterations on FactorGraph and on every rank of DistFactorGraph"""
    graph.add_neighborhood_factors(0, NF, r=2)
    if lowmem:
        graph.update_lowmem(steps=2, corr="alt")
        graph.add_factors([0, 6, 1, 5], [4, 2, 5, 1])             # later edges go to the owner of their source frame
        graph.rm_factors(graph.age > 1, store=True)               # a GLOBAL mask: the old edges become inactive on their owners
        graph.update_lowmem(steps=1, use_inactive=True, corr="alt")
    else:
        graph.update(1, use_inactive=True)
        graph.add_factors([0, 6, 1, 5], [4, 2, 5, 1])
        graph.rm_factors(graph.ii < 2, store=True)
        graph.update(None, None, use_inactive=True)               # t0 / t1 from the GLOBAL lists (the window is wider than any shard)
        graph.update(2, NF, motion_only=True)


def _state(graph, v):
    return dict(poses=v.poses[:NF].numpy().copy(), disps=v.disps[:NF].numpy().copy(), ii=graph.ii.numpy().copy(), jj=graph.jj.numpy().copy(),
                age=graph.age.numpy().copy(), ii_inac=graph.ii_inac.numpy().copy(), jj_inac=graph.jj_inac.numpy().copy())


class OracleVideoBA:
    """droid_backends.ba / ba_ex for the single-process run: the oracle's ba on the video's buffers"""

    def __init__(self, alpha=None):
        self.alpha = alpha

    def __call__(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, itrs, lm, ep, motion_only):
        from oracle import ba as oba
        p = poses.numpy().astype(np.float64); d = np.array(disps.numpy(), dtype=np.float64, order="C")
        if motion_only:
            K = len(np.unique(np.concatenate([np.arange(t0, t1), ii.numpy()])))
            eta = torch.full((K,) + tuple(disps.shape[1:]), 1e-3)
        oba.ba(p, d, intrinsics.numpy(), disps_sens.numpy(), targets.numpy(), weights.numpy(), eta.numpy(), ii.numpy(), jj.numpy(),
               t0, t1, itrs, lm, ep, motion_only, alpha_map=self.alpha)
        poses.copy_(torch.from_numpy(p).float()); disps.copy_(torch.from_numpy(d).float())


def _single(g, lowmem, with_conf):
    _install_fakes()
    from droid_amd import depth_video as dv_mod
    from droid_amd.factor_graph import FactorGraph
    v = _make_video(g, with_conf)
    ba = OracleVideoBA(g["conf"] if with_conf else None)
    fake = type("DB", (FakeDB,), {})
    fake.ba = staticmethod(ba)
    fake.ba_ex = staticmethod(lambda poses, disps, intr, sens, alpha, *rest: ba(poses, disps, intr, sens, *rest))
    dv_mod.db = fake
    graph = FactorGraph(v, FakeUpdate(), device="cpu", corr_impl="alt" if lowmem else "volume", native_corr=False)
    _script(graph, lowmem)
    return graph, v


def _worker(rank, world, port, out, lowmem, with_conf):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _install_fakes()
        from droid_amd.dist_ba import DistBA
        from droid_amd.dist_graph import DistFactorGraph
        from test_dist_cpu import OracleBackend

        class Backend(OracleBackend):
            """+ the motion-only build and the per-pixel prior of ba_build_shard_ex"""
            alpha_seen = False

            def ba_build(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, motion_only, alpha=None):
                from oracle import ba as oba
                p = poses.numpy().astype(np.float64).copy(); d = np.array(disps.numpy(), dtype=np.float64, order="C")
                e = eta.numpy()
                if motion_only:
                    K = len(np.unique(np.concatenate([np.arange(t0, t1), ii.numpy()])))
                    e = np.full((K,) + tuple(disps.shape[1:]), 1e-3, dtype=np.float32)
                _, _, info = oba.ba(p, d, intrinsics.numpy(), disps_sens.numpy(), targets.numpy(), weights.numpy(), e, ii.numpy(), jj.numpy(),
                                    t0, t1, 1, 0.0, 1e30, bool(motion_only), return_system=True, alpha_map=alpha)
                n = 6 * (t1 - t0)
                system = torch.zeros(n + 1, n, dtype=torch.float64)
                system[:n] = torch.from_numpy(info["H"]); system[n] = torch.from_numpy(info["b"])
                return dict(info=info, system=system, t0=t0, t1=t1, motion_only=bool(motion_only)), system

            def ba_build_shard_ex(self, poses, disps, intrinsics, disps_sens, alpha, *rest):
                Backend.alpha_seen = True
                return self.ba_build(poses, disps, intrinsics, disps_sens, *rest, alpha=alpha.numpy())

            def ba_finish_owned(self, poses, disps, jj, ws, n_eta_rows, t0, t1, lm, ep, motion_only, own_lo, own_hi):
                if not motion_only:
                    return super().ba_finish_owned(poses, disps, jj, ws, n_eta_rows, t0, t1, lm, ep, motion_only, own_lo, own_hi)
                from oracle import ba as oba
                system, n, P = ws["system"], 6 * (t1 - t0), t1 - t0
                if ws.get("flag"):
                    return torch.zeros(P, 6, dtype=torch.float64), None
                x, ok = oba.solve_damped(system[:n].numpy().copy(), system[n].numpy().copy(), lm, ep)
                dx = x.reshape(P, 6).astype(np.float32).astype(np.float64)
                pn = poses.numpy().astype(np.float64)
                oba._retract_poses(pn, dx, t0, t1, np.float64)
                poses.copy_(torch.from_numpy(pn).to(poses.dtype))
                return torch.from_numpy(dx), None

        g = _scenario()
        v = _make_video(g, with_conf)
        graph = DistFactorGraph(v, FakeUpdate(), device="cpu", corr_impl="alt" if lowmem else "volume", native_corr=False,
                                solver=DistBA(world, backend=Backend()))
        _script(graph, lowmem)
        # per-edge state lives on exactly one rank, in the order of the global list
        own = graph._owned(graph.ii)
        assert torch.equal(graph._lii, graph.ii[own]) and torch.equal(graph._ljj, graph.jj[own])
        assert graph._net.shape[0] == graph.target.shape[1] == graph.weight.shape[1] == int(own.sum())
        own_in = graph._owned(graph.ii_inac)
        assert torch.equal(graph._lii_inac, graph.ii_inac[own_in]) and graph.target_inac.shape[1] == int(own_in.sum())
        counts = torch.tensor([int(own.sum()), int(own_in.sum())]); dist.all_reduce(counts)
        assert counts.tolist() == [len(graph.ii), len(graph.ii_inac)]
        assert 0 < int(own.sum()) < len(graph.ii)                              # really sharded
        assert Backend.alpha_seen == bool(with_conf)
        # every rank ends with the same poses / depths and the same replicated lists
        chk = torch.cat([v.poses.flatten(), v.disps.flatten(), graph.ii.float(), graph.jj.float(), graph.age.float()])
        ref = chk.clone(); dist.broadcast(ref, 0)
        assert torch.equal(chk, ref)
        st = _state(graph, v)
        st["net_local"] = graph._net.numpy(); st["local_index"] = graph.local_index().numpy()
        st["target_local"] = graph.target[0].numpy()
        np.savez(out % rank, **st)
        graph.clear_edges()
        assert len(graph.ii) == 0 and len(graph._lii) == 0 and graph._net is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lowmem,with_conf", [(False, False), (True, False), (True, True)])
def test_sharded_graph_two_ranks_equals_single_process_graph(tmp_path, lowmem, with_conf):
    out = str(tmp_path / "rank%d.npz")
    port = 29300 + (os.getpid() % 2000) + 3 * int(lowmem) + 5 * int(with_conf)
    mp.spawn(_worker, args=(2, port, out, lowmem, with_conf), nprocs=2, join=True)
    g = _scenario()
    graph, v = _single(g, lowmem, with_conf)
    want = _state(graph, v)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    for k in ("ii", "jj", "age", "ii_inac", "jj_inac"):
        assert np.array_equal(r0[k], want[k]), k
    assert np.abs(r0["poses"] - want["poses"]).max() < 2e-6
    assert np.abs(r0["disps"] - want["disps"]).max() < 2e-5 * max(1.0, np.abs(want["disps"]).max())
    assert np.abs(want["poses"] - g["poses"]).max() > 1e-4                     # the iterations moved something
    # the ranks' hidden states / targets, put back into the global order, are the single-process ones
    net = np.zeros_like(graph._net.numpy()); tgt = np.zeros_like(graph.target[0].numpy())
    seen = np.zeros(len(want["ii"]), dtype=int)
    for r in (r0, r1):
        net[r["local_index"]] = r["net_local"]; tgt[r["local_index"]] = r["target_local"]; seen[r["local_index"]] += 1
    assert np.all(seen == 1)
    assert np.abs(net - graph._net.numpy()).max() < 1e-5 and np.abs(tgt - graph.target[0].numpy()).max() < 1e-4


def test_global_context_sums_follow_the_state_tensor_they_were_reduced_from():
    """host logic of round 6's chain (FactorGraph._operator): the sums the update operator leaves behind (last_glo) are handed to the next call
    only while the graph's hidden-state tensor IS the object they were reduced from -- not after rm_factors / add_factors (new tensor),
    not after a caller took the reference-shaped `net` view (it may write through it), never for update_lowmem's chunk copies."""
    _install_fakes()
    from droid_amd import depth_video as dv_mod
    from droid_amd.factor_graph import FactorGraph
    g = _scenario()
    v = _make_video(g, False)
    fake = type("DB", (FakeDB,), {})
    fake.ba = staticmethod(OracleVideoBA(None))
    dv_mod.db = fake
    calls = []

    class ChainUpdate(FakeUpdate):
        last_glo = None

        def fuses_next_glo(self, h, w):
            return True

        def forward_nhwc(self, net, inp, feats, flow, ii, inp_frames=None, inp_index=None, ctx=None, corr0=None, glo_red=None, glo_next=False,
                         want_upmask=True):
            calls.append((net, glo_red, glo_next, want_upmask))
            out = FakeUpdate.forward_nhwc(self, net, inp, feats, flow, ii, inp_frames=inp_frames, inp_index=inp_index, ctx=ctx, corr0=corr0)
            self.last_glo = net.float().sum((1, 2)) if glo_next else None      # "sums of the state this call wrote"
            return out
    graph = FactorGraph(v, ChainUpdate(), device="cpu", corr_impl="volume", native_corr=False)
    graph.add_neighborhood_factors(0, NF, r=2)
    graph.update(1)
    graph.update(1)
    graph.update(1)
    assert calls[0][1] is None and all(c[2] for c in calls)
    assert not any(c[3] for c in calls)                           # a graph that does not upsample does not ask for the upmask head ...
    graph.compute_upmask = True
    graph.update(1)
    assert calls.pop()[3]                                         # ... unless told to (bench.py)
    graph.compute_upmask = None
    assert calls[1][1] is not None and tuple(calls[1][1].shape) == (calls[0][0].shape[0], 16)            # what call 0 left behind
    assert calls[2][1] is not calls[1][1] and calls[2][0] is calls[1][0] is graph._net
    graph.rm_factors(graph.ii < 1, store=False)                   # the state tensor is replaced: nothing to hand over
    graph.update(1)
    assert calls[3][1] is None and calls[3][0] is graph._net and calls[3][0] is not calls[2][0]
    graph.update(1)
    assert calls[4][1] is not None
    _ = graph.net                                                  # reference-shaped view handed out: the caller may write through it
    graph.update(1)
    assert calls[5][1] is None
    graph.add_factors([0, 6], [4, 2])
    graph.update(1)
    assert calls[6][1] is None
    # update_lowmem's alt-correlation loop works on chunk COPIES of the state: never chained, and the kept sums are dropped
    n = len(calls)
    g2 = FactorGraph(v, ChainUpdate(), device="cpu", corr_impl="alt", native_corr=False)
    g2.add_neighborhood_factors(0, NF, r=2)
    g2.update_lowmem(steps=2, corr="alt")
    assert len(calls) > n and all(c[1] is None and not c[2] for c in calls[n:]) and g2._glo is None
