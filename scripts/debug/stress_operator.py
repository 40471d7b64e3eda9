"""diagnostic: N processes on one GPU, each repeating the same lookup + update operator call on its own fixed inputs; every repeat is
compared with the process's first result.  A timing-dependent race shows as sporadic mismatches that a process running alone never has."""
import os, sys, json
import numpy as np, torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "droid-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def worker(rank, nproc, reps, E, opts):
    import droid_backends as db
    import test_dist_graph_gpu as T
    from droid_amd.factor_graph import FactorGraph
    for k, val in opts.items():
        db.set_option(k, val)
    g = T._graph("C2")
    order = np.arange(len(g["ii"]))[:E]
    v, graph = T._setup(g, FactorGraph, "volume", False, order=order)
    graph.upsample = False
    ii, jj = graph.ii, graph.jj
    coords1 = v.reproject(ii, jj)[0][0]
    net0 = graph._net.clone()
    tprev = graph.target[0].contiguous()
    ref = None
    nbad = {"net": 0, "delta": 0, "weight": 0, "damping": 0, "corr0": 0}
    worst = {"net": 0.0, "delta": 0.0, "weight": 0.0, "damping": 0.0, "corr0": 0.0}
    for it in range(reps):
        graph._net.copy_(net0); graph._glo = None
        feats, corr0 = graph._pyramid_features(graph.corr, coords1)
        c0 = (corr0 if corr0 is not None else feats).float().clone()
        dw, damping, upmask, uniq = graph._operator(graph._net, coords1, tprev, feats, ii, corr0)
        torch.cuda.synchronize()
        cur = {"net": graph._net.float().clone(), "delta": dw[..., :2].clone(), "weight": dw[..., 2:].clone(), "damping": damping.clone(), "corr0": c0}
        if ref is None:
            ref = cur
            continue
        tol = {"net": 2.0 ** -8, "delta": 2e-3, "weight": 2e-3, "damping": 1e-6, "corr0": 2.0 ** -6}
        for k in cur:
            dmax = float((cur[k] - ref[k]).abs().max())
            worst[k] = max(worst[k], dmax)
            if dmax > tol[k]:
                nbad[k] += 1
                if nbad[k] <= 2:
                    bad = torch.nonzero((cur[k] - ref[k]).abs() > tol[k])
                    print("  proc %d it %d %s: %d elements off, max %.3g, first %s last %s" % (rank, it, k, len(bad), dmax, bad[0].tolist(), bad[-1].tolist()), flush=True)
    print("proc %d/%d E=%d %s: mismatching repeats of %d: %s; worst %s" % (rank, nproc, E, opts, reps - 1, nbad, {k: "%.3g" % x for k, x in worst.items()}), flush=True)


if __name__ == "__main__":
    nproc, reps, E = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    opts = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
    mp.spawn(worker, args=(nproc, reps, E, opts), nprocs=nproc, join=True)
