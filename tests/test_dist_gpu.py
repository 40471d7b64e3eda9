"""GPU: the edge-sharded BA (droid_amd/dist_ba.py) with the REAL kernels under a 2-rank process group.  One MI355X per box
here, and RCCL refuses two ranks on one device, so the ranks share cuda:0 and talk through gloo: everything but the
transport (partition, dh_ba_build per shard, packed all-reduce of the co-visible blocks, redundant dh_ba_finish, owner-only
depth updates, final depth exchange) is what runs on a multi-GPU node."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "droid-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _worker(rank, world, port, out, cfg, packed):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from droid_amd import synthetic as syn
        from droid_amd.dist_ba import DistBA, shard_edges_by_source_frame, local_eta_rows
        g = syn.make_graph(cfg)
        N, t0, t1 = g["n_frames"], 1, g["n_frames"]
        shards, bounds = shard_edges_by_source_frame(g["ii"], world)
        mine = shards[rank]
        rows, _ = local_eta_rows(g["ii"], g["ii"][mine], t0, t1)
        d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
        poses, disps = d(g["poses"]), d(g["disps"])
        solver = DistBA(world)
        solver.set_owned_frames(bounds[rank], bounds[rank + 1])
        if packed:
            solver.set_graph(g["ii"], g["jj"], t0, t1)
        solver.ba(poses, disps, d(g["intrinsics"]), d(g["disps_sens"]), d(g["targets"][mine]), d(g["weights"][mine]),
                  d(g["eta"][rows]), d(g["ii"][mine]), d(g["jj"][mine]), t0, t1, 2, g["lm"], g["ep"])
        torch.cuda.synchronize()
        chk = torch.cat([poses.flatten(), disps.flatten()]).cpu()
        ref = chk.clone(); dist.broadcast(ref, 0)
        assert torch.allclose(chk, ref, atol=1e-6)                 # every rank ends with the same state
        if rank == 0:
            np.savez(out, poses=poses.cpu().numpy(), disps=disps.cpu().numpy(), exchange_bytes=solver.last_exchange_bytes)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg,packed", [("C1", True), ("C2", True), ("C2", False)])
def test_two_rank_sharded_ba_with_real_kernels_equals_single_gpu(tmp_path, cfg, packed):
    assert torch.cuda.is_available()
    import droid_backends as db
    from droid_amd import synthetic as syn
    out = str(tmp_path / "rank0.npz")
    port = 29600 + (os.getpid() % 1500) + (11 if packed else 0) + (23 if cfg == "C2" else 0)
    mp.spawn(_worker, args=(2, port, out, cfg, packed), nprocs=2, join=True)
    got = np.load(out)
    g = syn.make_graph(cfg)
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    poses, disps = d(g["poses"]), d(g["disps"])
    db.ba(poses, disps, d(g["intrinsics"]), d(g["disps_sens"]), d(g["targets"]), d(g["weights"]), d(g["eta"]), d(g["ii"]), d(g["jj"]),
          1, g["n_frames"], 2, g["lm"], g["ep"], False)
    torch.cuda.synchronize()
    assert np.abs(got["poses"] - poses.cpu().numpy()).max() < 2e-5
    rd = disps.cpu().numpy()
    e = np.abs(got["disps"] - rd) / np.maximum(1.0, np.abs(rd))
    assert np.quantile(e, 0.995) <= 1e-4 and e.max() <= 1e-2
    n = 6 * (g["n_frames"] - 1)
    npad = (n + 63) // 64 * 64
    dense = (npad + 64) * npad * 8
    assert int(got["exchange_bytes"]) == dense if not packed else int(got["exchange_bytes"]) < dense // 2


# ------------------------------------------------------------------------------------------ RCCL, one rank
def _nccl_worker(rank, world, port, out, cfg):
    """A group of ONE rank on the `nccl` backend (= RCCL on ROCm): librccl is loaded, a communicator is created on cuda:0 and
    the packed all-reduce, the status words and the final depth all-reduce of DistBA really go through it."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from droid_amd import synthetic as syn
        from droid_amd.dist_ba import DistBA
        g = syn.make_graph(cfg)
        t0, t1 = 1, g["n_frames"]
        d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
        poses, disps = d(g["poses"]), d(g["disps"])
        probe = torch.ones(4, device="cuda"); dist.all_reduce(probe); torch.cuda.synchronize()       # the communicator exists
        assert dist.get_backend() == "nccl" and float(probe.sum()) == 4.0
        solver = DistBA(world, always_reduce=True)
        solver.set_owned_frames(0, 1 << 30)
        solver.set_graph(g["ii"], g["jj"], t0, t1)
        solver.ba(poses, disps, d(g["intrinsics"]), d(g["disps_sens"]), d(g["targets"]), d(g["weights"]), d(g["eta"]),
                  d(g["ii"]), d(g["jj"]), t0, t1, 2, g["lm"], g["ep"])
        torch.cuda.synchronize()
        np.savez(out, poses=poses.cpu().numpy(), disps=disps.cpu().numpy(), exchange_bytes=solver.last_exchange_bytes,
                 packed=solver.last_exchange_packed)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg", ["C1", "C2"])
def test_one_rank_nccl_group_runs_the_packed_exchange_through_rccl(tmp_path, cfg):
    """world-size-1 `nccl` process group on the box's one GPU: RCCL loaded, communicator created, the packed all-reduce
    executed on the device buffer the pack kernel wrote; with one rank the sum is the identity and the pack / unpack
    round trip is exact, so the result must EQUAL droid_backends.ba bit for bit."""
    assert torch.cuda.is_available()
    import droid_backends as db
    from droid_amd import synthetic as syn
    out = str(tmp_path / "nccl.npz")
    port = 29900 + (os.getpid() % 1500) + (31 if cfg == "C2" else 0)
    mp.spawn(_nccl_worker, args=(1, port, out, cfg), nprocs=1, join=True)
    got = np.load(out)
    assert bool(got["packed"])
    g = syn.make_graph(cfg)
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    poses, disps = d(g["poses"]), d(g["disps"])
    db.ba(poses, disps, d(g["intrinsics"]), d(g["disps_sens"]), d(g["targets"]), d(g["weights"]), d(g["eta"]), d(g["ii"]), d(g["jj"]),
          1, g["n_frames"], 2, g["lm"], g["ep"], False)
    torch.cuda.synchronize()
    assert np.array_equal(got["poses"], poses.cpu().numpy()) and np.array_equal(got["disps"], disps.cpu().numpy())
    n = 6 * (g["n_frames"] - 1)
    assert int(got["exchange_bytes"]) < ((n + 63) // 64 * 64 + 64) * ((n + 63) // 64 * 64) * 8


# ------------------------------------------------------------------------------------------ errors are collective
def _bad_worker(rank, world, port, out, what):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from droid_amd import synthetic as syn
        from droid_amd.dist_ba import DistBA, shard_edges_by_source_frame, local_eta_rows
        g = syn.make_graph("C1")
        t0, t1 = 1, g["n_frames"]
        shards, bounds = shard_edges_by_source_frame(g["ii"], world)
        mine = shards[rank]
        rows, _ = local_eta_rows(g["ii"], g["ii"][mine], t0, t1)
        d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
        jj = g["jj"][mine].copy()
        pat_ii, pat_jj = g["ii"], g["jj"]
        if what == "bad_index" and rank == 1:
            jj[0] = 1000                                          # outside the frame buffer, on ONE rank only
        if what == "stale_pattern":
            keep = np.abs(g["ii"] - g["jj"]) <= 1                 # pattern of an older, smaller edge list
            pat_ii, pat_jj = g["ii"][keep], g["jj"][keep]
        poses, disps = d(g["poses"]), d(g["disps"])
        solver = DistBA(world)
        solver.set_owned_frames(bounds[rank], bounds[rank + 1])
        solver.set_graph(pat_ii, pat_jj, t0, t1)
        raised = False
        try:
            solver.ba(poses, disps, d(g["intrinsics"]), d(g["disps_sens"]), d(g["targets"][mine]), d(g["weights"][mine]),
                      d(g["eta"][rows]), d(g["ii"][mine]), d(jj), t0, t1, 2, g["lm"], g["ep"])
        except RuntimeError:
            raised = True
        torch.cuda.synchronize()
        np.savez(out % rank, raised=raised, poses=poses.cpu().numpy(), disps=disps.cpu().numpy(), packed=solver.last_exchange_packed)
    finally:
        dist.destroy_process_group()


def test_bad_edge_index_on_one_rank_raises_on_every_rank_and_applies_no_update(tmp_path):
    """a rank-local strict check ahead of the all-reduce would raise on one rank and leave the other waiting in
    the collective.  The flag now travels with the exchanged buffer: BOTH ranks leave ba() with a RuntimeError and
    untouched poses / depths."""
    from droid_amd import synthetic as syn
    out = str(tmp_path / "bad%d.npz")
    mp.spawn(_bad_worker, args=(2, 29750 + (os.getpid() % 1500), out, "bad_index"), nprocs=2, join=True)
    g = syn.make_graph("C1")
    for r in (0, 1):
        got = np.load(out % r)
        assert bool(got["raised"])
        assert np.array_equal(got["poses"], g["poses"]) and np.array_equal(got["disps"], g["disps"])


def test_stale_block_pattern_falls_back_to_the_dense_exchange_on_every_rank(tmp_path):
    from droid_amd import synthetic as syn
    import droid_backends as db
    out = str(tmp_path / "stale%d.npz")
    mp.spawn(_bad_worker, args=(2, 29850 + (os.getpid() % 1500), out, "stale_pattern"), nprocs=2, join=True)
    g = syn.make_graph("C1")
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    poses, disps = d(g["poses"]), d(g["disps"])
    db.ba(poses, disps, d(g["intrinsics"]), d(g["disps_sens"]), d(g["targets"]), d(g["weights"]), d(g["eta"]), d(g["ii"]), d(g["jj"]),
          1, g["n_frames"], 2, g["lm"], g["ep"], False)
    torch.cuda.synchronize()
    for r in (0, 1):
        got = np.load(out % r)
        assert not bool(got["raised"]) and not bool(got["packed"])
        assert np.abs(got["poses"] - poses.cpu().numpy()).max() < 2e-5
        e = np.abs(got["disps"] - disps.cpu().numpy()) / np.maximum(1.0, np.abs(disps.cpu().numpy()))
        assert np.quantile(e, 0.995) <= 1e-4 and e.max() <= 1e-2
