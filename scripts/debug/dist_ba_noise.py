"""diagnostic: (A) two-rank DistBA vs db.ba on fixed inputs at two dampings; (C) single-process graph run while another process keeps the GPU busy"""
import os, sys, json, time
import numpy as np, torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "droid-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def ba_worker(rank, world, port, out, lm, ep):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from droid_amd import synthetic as syn
    from droid_amd.dist_ba import DistBA, shard_edges_by_source_frame, local_eta_rows
    g = syn.make_graph("C2")
    N, t0, t1 = g["n_frames"], 1, g["n_frames"]
    shards, bounds = shard_edges_by_source_frame(g["ii"], world)
    mine = shards[rank]
    rows, _ = local_eta_rows(g["ii"], g["ii"][mine], t0, t1)
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    poses, disps = d(g["poses"]), d(g["disps"])
    solver = DistBA(world)
    solver.set_owned_frames(bounds[rank], bounds[rank + 1])
    solver.set_graph(g["ii"], g["jj"], t0, t1)
    solver.ba(poses, disps, d(g["intrinsics"]), d(g["disps_sens"]), d(g["targets"][mine]), d(g["weights"][mine]),
              d(g["eta"][rows]), d(g["ii"][mine]), d(g["jj"][mine]), t0, t1, 2, lm, ep)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, poses=poses.cpu().numpy(), disps=disps.cpu().numpy())
    dist.destroy_process_group()


def busy(stop):
    x = torch.randn(4096, 4096, device="cuda")
    while not stop.is_set():
        y = x @ x
        torch.cuda.synchronize()


def main():
    import droid_backends as db
    from droid_amd import synthetic as syn
    what = sys.argv[1]
    if what == "ba":
        for lm, ep in ((1e-4, 0.1), (1e-5, 1e-2)):
            g = syn.make_graph("C2")
            d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
            refs = []
            for k in range(2):
                poses, disps = d(g["poses"]), d(g["disps"])
                db.ba(poses, disps, d(g["intrinsics"]), d(g["disps_sens"]), d(g["targets"]), d(g["weights"]), d(g["eta"]), d(g["ii"]), d(g["jj"]),
                      1, g["n_frames"], 2, lm, ep, False)
                refs.append(poses.cpu().numpy())
            print("lm %g ep %g single run-to-run poses %.3g" % (lm, ep, np.abs(refs[0] - refs[1]).max()))
            for k in range(4):
                out = "/tmp/dbn_%d.npz" % k
                mp.spawn(ba_worker, args=(2, 29450 + k, out, lm, ep), nprocs=2, join=True)
                print("  DistBA 2 ranks rep %d: poses %.3g" % (k, np.abs(np.load(out)["poses"] - refs[0]).max()), flush=True)
    else:
        import test_dist_graph_gpu as T
        from droid_amd.factor_graph import FactorGraph
        g = T._graph("C2"); N = g["n_frames"]
        def one():
            v, graph = T._setup(g, FactorGraph, "alt", False)
            graph.upsample = False
            graph.update_lowmem(steps=2, corr="pyramid")
            torch.cuda.synchronize()
            return v.poses[:N].cpu().numpy()
        ref = one()
        print("alone: run-to-run %.3g" % np.abs(one() - ref).max())
        ctx = mp.get_context("spawn")
        stop = ctx.Event()
        p = ctx.Process(target=busy, args=(stop,)); p.start()
        time.sleep(5)
        for k in range(4):
            print("with a busy neighbour process rep %d: %.3g" % (k, np.abs(one() - ref).max()), flush=True)
        stop.set(); p.join()


if __name__ == "__main__":
    main()
