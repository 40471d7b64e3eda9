"""diagnostic: does a process's result change because ANOTHER process runs on the same GPU?
victim    = a loop of small kernels on fixed inputs, every result compared bit for bit with the first one
            ("ours": droid_backends.reproject; "torch": an elementwise chain with a reciprocal, torch kernels only)
aggressor = a second process looping "ours" (update_lowmem steps of a C2 shard: pyramid build, fused lookup, update operator) or
            "torch" (fp16 matmuls + conv2d + elementwise), for `secs` seconds."""
import os, sys, time
import numpy as np, torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "droid-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def aggressor(kind, go, secs):
    torch.cuda.set_device(0)
    if kind == "ours":
        import test_dist_graph_gpu as T
        from droid_amd.factor_graph import FactorGraph
        g = T._graph("C2")
        v, graph = T._setup(g, FactorGraph, "alt", False, order=np.arange(259))
        graph.upsample = False
        graph._solve = lambda *a, **k: None
        step = lambda: graph.update_lowmem(steps=1, corr="pyramid")
    elif kind in ("build", "lookup", "operator", "reproject", "ba"):
        import droid_backends as db
        import test_dist_graph_gpu as T
        from droid_amd.factor_graph import FactorGraph
        from droid_amd.corr import CorrBlock
        g = T._graph("C2")
        v, graph = T._setup(g, FactorGraph, "volume", False, order=np.arange(259))
        graph.upsample = False
        ii, jj = graph.ii, graph.jj
        coords1 = v.reproject(ii, jj)[0][0]
        feats, corr0 = graph._pyramid_features(graph.corr, coords1)
        tprev = graph.target[0].contiguous()
        arena = CorrBlock.arena(259, 48, 64, "cuda")
        d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
        s2 = __import__("droid_amd.synthetic", fromlist=["x"]).make_graph("C2")
        bargs = [d(s2[k]) for k in ("intrinsics", "disps_sens", "targets", "weights", "eta", "ii", "jj")]
        p0, d0 = d(s2["poses"]), d(s2["disps"])
        step = {"build": lambda: CorrBlock.from_frames(v.fmaps, ii, jj, out=arena),
                "lookup": lambda: graph._pyramid_features(graph.corr, coords1),
                "operator": lambda: graph._operator(graph._net, coords1, tprev, feats, ii, corr0),
                "reproject": lambda: [v.reproject(ii, jj) for _ in range(50)],
                "ba": lambda: db.ba(p0.clone(), d0.clone(), *bargs, 1, 64, 2, 1e-4, 0.1, False)}[kind]
    elif kind == "torch":
        a = torch.randn(4096, 4096, device="cuda", dtype=torch.half)
        x = torch.randn(64, 128, 48, 64, device="cuda", dtype=torch.half)
        w = torch.randn(128, 128, 3, 3, device="cuda", dtype=torch.half)
        def step():
            (a @ a); torch.nn.functional.conv2d(x, w, padding=1); torch.sigmoid(x) * x
    else:
        return
    step(); torch.cuda.synchronize()
    while time.time() < go:
        pass
    n = 0
    while time.time() < go + secs:
        step(); n += 1
    torch.cuda.synchronize()
    print("aggressor (%s): %d steps" % (kind, n), flush=True)


def victim(kind, go, secs):
    torch.cuda.set_device(0)
    if kind == "ours":
        import droid_backends as db
        import test_dist_graph_gpu as T
        g = T._graph("C2")
        d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
        poses, disps, intr = d(g["poses"]), d(g["disps"]), d(np.tile(g["intrinsics"], (g["n_frames"], 1)))
        ii, jj = d(g["ii"][:259]), d(g["jj"][:259])
        run = lambda: db.reproject(poses, disps, intr, ii, jj)[0]
    else:
        x = torch.rand(259, 48, 64, 2, device="cuda") + 0.5
        run = lambda: (x * 1.0001 + 2.0).reciprocal() * x + (x * x - 0.25)
    ref = run().clone(); torch.cuda.synchronize()
    while time.time() < go:
        pass
    n = bad = 0
    while time.time() < go + secs:
        outs = [run() for _ in range(16)]
        for o in outs:
            nb = int((o != ref).sum())
            if nb:
                bad += 1
                if bad <= 3:
                    ix = torch.nonzero(o != ref)
                    print("  victim (%s): %d elements differ; first %s last %s; max |diff| %.3g" % (kind, nb, ix[0].tolist(), ix[-1].tolist(), float((o - ref).abs().max())), flush=True)
        n += 16
    print("victim (%s): %d of %d results differ from the first one" % (kind, bad, n), flush=True)


def main():
    vk, ak, secs = sys.argv[1], sys.argv[2], float(sys.argv[3])
    go = time.time() + 25
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=victim, args=(vk, go, secs)), ctx.Process(target=aggressor, args=(ak, go, secs))]
    for p in ps: p.start()
    for p in ps: p.join()


if __name__ == "__main__":
    main()
