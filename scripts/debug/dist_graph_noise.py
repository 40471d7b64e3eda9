"""diagnostic: run-to-run spread of the two-rank sharded graph against the single-process graph (one GPU, gloo)"""
import os, sys, json
import numpy as np, torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "droid-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_dist_graph_gpu as T


def worker(rank, world, port, out, name, mode, conf, opts):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from droid_amd.dist_graph import DistFactorGraph
    g = T._graph(name)
    v, graph = T._setup(g, DistFactorGraph, "alt" if mode == "lowmem" else "volume", conf)
    graph.upsample = opts.get("upsample", True)
    if opts.get("dense"):
        graph._set_pattern = lambda *a: None
    run(graph, mode, opts)
    N = g["n_frames"]
    np.savez(out % rank, poses=v.poses[:N].cpu().numpy(), disps=v.disps[:N].cpu().numpy())
    dist.destroy_process_group()


def run(graph, mode, opts):
    if mode == "lowmem":
        graph.update_lowmem(steps=opts.get("steps", 2), corr=opts.get("corr", "auto"))
    else:
        for _ in range(opts.get("steps", 2)):
            graph.update(1, None, itrs=2, use_inactive=False, lm=opts.get("lm", 1e-4), ep=opts.get("ep", 0.1))
    torch.cuda.synchronize()


def main():
    from droid_amd.factor_graph import FactorGraph
    name, mode, conf = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
    opts = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
    g = T._graph(name); N = g["n_frames"]
    ref = []
    for k in range(2):
        v, graph = T._setup(g, FactorGraph, "alt" if mode == "lowmem" else "volume", conf)
        graph.upsample = opts.get("upsample", True)
        run(graph, mode, opts)
        ref.append((v.poses[:N].cpu().numpy(), v.disps[:N].cpu().numpy()))
    print("single run-to-run: poses %.3g disps %.3g" % (np.abs(ref[0][0] - ref[1][0]).max(), np.abs(ref[0][1] - ref[1][1]).max()))
    for k in range(opts.get("reps", 4)):
        out = "/tmp/dgn_%d_%%d.npz" % k
        mp.spawn(worker, args=(2, 29400 + k, out, name, mode, conf, opts), nprocs=2, join=True)
        r = np.load(out % 0)
        e = np.abs(r["disps"] - ref[0][1]) / np.maximum(1, np.abs(ref[0][1]))
        print("%s %s conf=%d %s rep %d: poses %.3g disps q99 %.3g max %.3g" % (name, mode, conf, opts, k, np.abs(r["poses"] - ref[0][0]).max(), np.quantile(e, 0.99), e.max()), flush=True)


if __name__ == "__main__":
    main()
