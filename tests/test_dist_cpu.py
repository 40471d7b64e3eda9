"""world_size-2 gloo test of the edge-sharded BA host logic (droid_amd/dist_ba.py) on CPU.

The per-rank numerics (dh_ba_build_shard / dh_ba_pack_blocks / dh_ba_unpack_blocks / dh_ba_finish_owned) need a GPU; here they are replaced by an injected backend that
computes the same two halves with the oracle, so what is exercised is exactly what runs between the kernels on
a multi-GPU node: the partition by source frame, the eta row mapping, the all-reduce of the reduced camera system,
owner-only depth updates and the final exchange of depth increments.  Sharded result == unsharded oracle.
"""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "droid-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


class OracleBackend:
    """ba_build / ba_finish with the contract of droid_backends (CPU tensors, oracle arithmetic, fp64)."""

    def ba_build(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, motion_only):
        from oracle import ba as oba
        assert not motion_only
        p = poses.numpy().astype(np.float64).copy(); d = disps.numpy().astype(np.float64).copy()
        _, _, info = oba.ba(p, d, intrinsics.numpy(), disps_sens.numpy(), targets.numpy(), weights.numpy(), eta.numpy(),
                            ii.numpy(), jj.numpy(), t0, t1, 1, 0.0, 1e30, False, return_system=True)
        n = 6 * (t1 - t0)
        system = torch.zeros(n + 1, n, dtype=torch.float64)
        system[:n] = torch.from_numpy(info["H"]); system[n] = torch.from_numpy(info["b"])
        return dict(info=info, system=system, t0=t0, t1=t1), system

    def ba_build_shard(self, *a):                      # droid_backends.ba_build_shard: ba_build without the host check
        return self.ba_build(*a)

    @staticmethod
    def _block_index(bp, bq):
        r = torch.arange(6)
        return 6 * bp.long()[:, None, None] + r[None, :, None], 6 * bq.long()[:, None, None] + r[None, None, :]

    def ba_pack_blocks(self, ws, disps, jj, t0, t1, motion_only, bp, bq, host_flags, packed):
        system, n, nb = ws["system"], 6 * (t1 - t0), len(bp)
        rows, cols = self._block_index(bp, bq)
        packed[:36 * nb] = system[rows, cols].reshape(-1)
        packed[36 * nb:36 * nb + n] = system[n]
        packed[36 * nb + n] = 0.0
        packed[36 * nb + n + 1] = float(host_flags != 0)

    def ba_unpack_blocks(self, ws, disps, jj, t0, t1, motion_only, bp, bq, packed):
        system, n, nb = ws["system"], 6 * (t1 - t0), len(bp)
        rows, cols = self._block_index(bp, bq)
        system[rows, cols] = packed[:36 * nb].reshape(nb, 6, 6)
        system[n] = packed[36 * nb:36 * nb + n]
        ws["flag"] = bool(packed[36 * nb + n] != 0 or packed[36 * nb + n + 1] != 0)

    def ba_exchange_flags(self, ws, disps, jj, t0, t1, motion_only, host_flags, status, set):
        if set:
            ws["flag"] = bool((status != 0).any())
        else:
            status[0] = 0.0; status[1] = float(host_flags != 0)

    def ba_finish_owned(self, poses, disps, jj, ws, n_eta_rows, t0, t1, lm, ep, motion_only, own_lo, own_hi):
        from oracle import ba as oba
        info, system = ws["info"], ws["system"]
        n = 6 * (t1 - t0)
        P = t1 - t0
        HW = info["Q"].shape[1]
        if ws.get("flag"):                                  # some rank flagged: nobody updates
            return torch.zeros(P, 6, dtype=torch.float64), torch.zeros(len(info["kx"]), HW, dtype=torch.float64)
        x, ok = oba.solve_damped(system[:n].numpy().copy(), system[n].numpy().copy(), lm, ep)
        dx = x.reshape(P, 6).astype(np.float32).astype(np.float64)
        prel = info["jj_exp"] - t0
        use = (prel > 0) & (prel < P)
        dw = np.zeros((len(prel), HW))
        dw[use] = np.einsum("nip,ni->np", info["Erow"][use], dx[prel[use]])
        dz = info["Q"] * (info["w"] - oba._segsum(dw, info["ii_exp"], info["kx"]))
        dz[(info["kx"] < own_lo) | (info["kx"] >= own_hi)] = 0.0          # only the owner of a frame moves its depths
        pn = poses.numpy().astype(np.float64)
        oba._retract_poses(pn, dx, t0, t1, np.float64)
        poses.copy_(torch.from_numpy(pn).to(poses.dtype))
        dflat = disps.view(disps.shape[0], -1)
        dflat[torch.from_numpy(info["kx"])] += torch.from_numpy(dz).to(disps.dtype)
        return torch.from_numpy(dx), torch.from_numpy(dz)


def _worker(rank, world, port, out, packed=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from droid_amd import synthetic as syn
        from droid_amd.dist_ba import DistBA, shard_edges_by_source_frame, local_eta_rows
        g = syn.small_graph(n_frames=7, seed=11, ht=8, wd=12, radius=3)
        N, t0, t1 = g["n_frames"], 1, g["n_frames"]
        shards, bounds = shard_edges_by_source_frame(g["ii"], world)
        mine = shards[rank]
        rows, _ = local_eta_rows(g["ii"], g["ii"][mine], t0, t1)
        T = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a)).to(dt)
        poses, disps = T(g["poses"]), T(g["disps"])
        solver = DistBA(world, backend=OracleBackend())
        solver.set_owned_frames(bounds[rank], bounds[rank + 1])
        if packed == "stale":
            # pattern of an OLDER, smaller edge list (what is left behind when the global edges change without a new
            # set_graph): the local blocks are not covered -> every rank must fall back to the dense exchange
            keep = np.abs(g["ii"] - g["jj"]) <= 1
            solver.set_graph(g["ii"][keep], g["jj"][keep], t0, t1)
        elif packed:
            solver.set_graph(g["ii"], g["jj"], t0, t1)          # all-reduce of the co-visible 6x6 blocks only
        solver.ba(poses, disps, T(g["intrinsics"]), T(g["disps_sens"]), T(g["targets"][mine]), T(g["weights"][mine]),
                  T(g["eta"][rows]), T(g["ii"][mine], torch.int64), T(g["jj"][mine], torch.int64), t0, t1, 2, 1e-4, 0.1)
        if rank == 0:
            np.savez(out, poses=poses.numpy(), disps=disps.numpy(), exchange_bytes=solver.last_exchange_bytes,
                     packed=solver.last_exchange_packed)
        # every rank ends with the same state
        chk = torch.cat([poses.flatten(), disps.flatten()])
        ref = chk.clone(); dist.broadcast(ref, 0)
        assert torch.allclose(chk, ref, atol=1e-12)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("packed", [False, True, "stale"])
def test_sharded_ba_two_ranks_gloo(tmp_path, packed):
    from droid_amd import synthetic as syn
    from oracle import ba as oba
    out = str(tmp_path / "rank0.npz")
    port = 29500 + (os.getpid() % 2000) + {False: 0, True: 7, "stale": 13}[packed]
    mp.spawn(_worker, args=(2, port, out, packed), nprocs=2, join=True)
    got = np.load(out)
    n = 6 * 6
    if packed is True:
        assert bool(got["packed"]) and int(got["exchange_bytes"]) < (n + 1) * n * 8
    else:                                                   # no pattern, or a stale one that does not cover the edges
        assert not bool(got["packed"]) and int(got["exchange_bytes"]) == (n + 1) * n * 8
    g = syn.small_graph(n_frames=7, seed=11, ht=8, wd=12, radius=3)
    p = g["poses"].astype(np.float64).copy(); d = g["disps"].astype(np.float64).copy()
    oba.ba(p, d, g["intrinsics"], g["disps_sens"], g["targets"], g["weights"], g["eta"], g["ii"], g["jj"],
           1, g["n_frames"], 2, 1e-4, 0.1, False)
    assert np.abs(got["poses"] - p).max() < 1e-6          # dx is rounded to fp32 in both, summation order differs
    assert np.abs(got["disps"] - d).max() < 1e-5 * max(1.0, np.abs(d).max())


def _bad_partition_worker(rank, world, port, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from droid_amd import synthetic as syn
        from droid_amd.dist_ba import DistBA, shard_edges_by_source_frame, local_eta_rows
        g = syn.small_graph(n_frames=7, seed=11, ht=8, wd=12, radius=3)
        t0, t1 = 1, g["n_frames"]
        shards, bounds = shard_edges_by_source_frame(g["ii"], world)
        mine = shards[rank]
        rows, _ = local_eta_rows(g["ii"], g["ii"][mine], t0, t1)
        T = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a)).to(dt)
        poses, disps = T(g["poses"]), T(g["disps"])
        solver = DistBA(world, backend=OracleBackend())
        if mode == "overlap":                                  # frame bounds[1] owned by both ranks
            solver.set_owned_frames(bounds[rank], bounds[rank + 1] + (1 if rank == 0 else 0))
        elif mode == "gap":                                    # frame bounds[1] owned by nobody
            solver.set_owned_frames(bounds[rank] + (1 if rank == 1 else 0), bounds[rank + 1])
        msg = ""
        try:
            solver.ba(poses, disps, T(g["intrinsics"]), T(g["disps_sens"]), T(g["targets"][mine]), T(g["weights"][mine]),
                      T(g["eta"][rows]), T(g["ii"][mine], torch.int64), T(g["jj"][mine], torch.int64), t0, t1, 2, 1e-4, 0.1)
        except RuntimeError as exc:
            msg = str(exc)
        unchanged = bool(np.array_equal(disps.numpy(), g["disps"].astype(np.float64)))
        open("%s.%d" % (out, rank), "w").write("%d|%s" % (unchanged, msg))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["never_set", "overlap", "gap"])
def test_ownership_ranges_that_do_not_partition_the_frames_are_refused_on_every_rank(tmp_path, mode):
    """the final depth exchange zeroes the maps a rank does not own and SUMS: ranges that overlap, leave a gap, or were never
    set (the constructor default = everything) would multiply / zero depth maps silently -- every rank raises before anything moves"""
    out = str(tmp_path / "r")
    port = 29700 + (os.getpid() % 2000) + {"never_set": 0, "overlap": 5, "gap": 9}[mode]
    mp.spawn(_bad_partition_worker, args=(2, port, mode, out), nprocs=2, join=True)
    for rank in (0, 1):
        unchanged, msg = open("%s.%d" % (out, rank)).read().split("|", 1)
        assert unchanged == "1"
        assert ("set_owned_frames() was never called" in msg) if mode == "never_set" else ("do not partition" in msg)


def test_partition_covers_every_edge_once_and_keeps_source_frames_together():
    from droid_amd.dist_ba import shard_edges_by_source_frame
    rng = np.random.default_rng(0)
    ii = rng.integers(0, 50, 400)
    for world in (1, 2, 3, 8):
        shards, bounds = shard_edges_by_source_frame(ii, world)
        allidx = np.sort(np.concatenate(shards))
        assert np.array_equal(allidx, np.arange(len(ii)))
        owners = {}
        for r, s in enumerate(shards):
            for f in np.unique(ii[s]):
                assert owners.setdefault(int(f), r) == r
                assert bounds[r] <= f < bounds[r + 1]


def test_reduced_system_pattern_covers_the_nonzero_blocks():
    """every 6x6 block of the oracle's reduced camera system that is not identically zero lies in the pattern"""
    from droid_amd import synthetic as syn
    from droid_amd.dist_ba import reduced_system_pattern
    from oracle import ba as oba
    g = syn.small_graph(n_frames=9, seed=3, ht=8, wd=12, radius=2)
    t0, t1 = 2, 9
    p = g["poses"].astype(np.float64); d = np.array(g["disps"], dtype=np.float64, order="C")
    kx = np.unique(np.concatenate([np.arange(t0, t1), g["ii"]]))
    eta = np.full((len(kx),) + g["disps"].shape[1:], 1e-4, dtype=np.float32)
    _, _, info = oba.ba(p, d, g["intrinsics"], g["disps_sens"], g["targets"], g["weights"], eta, g["ii"], g["jj"], t0, t1, 1, 1e-4, 0.1, False, return_system=True)
    P = t1 - t0
    nz = np.abs(info["H"].reshape(P, 6, P, 6)).max(axis=(1, 3)) > 0
    pp, qq = reduced_system_pattern(g["ii"], g["jj"], t0, t1)
    pat = np.zeros((P, P), bool); pat[pp, qq] = True
    assert np.all(pp >= qq) and not np.any(np.tril(nz) & ~pat)
    assert pat.sum() < P * (P + 1) // 2                       # and it is sparse: not every pair is co-visible
