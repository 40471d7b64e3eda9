"""diagnostic: first update_lowmem step, two ranks vs single process: BA inputs per rank and BA outputs"""
import os, sys, json
import numpy as np, torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "droid-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_dist_graph_gpu as T


def instrument(graph, store):
    orig = graph._solve
    def solve(tb, wb, ii, jj, t0, t1, itrs, lm, ep, motion_only, EP, uniq=None):
        v = graph.video
        store["tb"] = tb.cpu().numpy(); store["wb"] = wb.cpu().numpy(); store["ii"] = ii.cpu().numpy(); store["jj"] = jj.cpu().numpy()
        store["damping"] = graph.damping.cpu().numpy()
        store["poses_in"] = v.poses.cpu().numpy(); store["disps_in"] = v.disps.cpu().numpy()
        try:
            orig(tb, wb, ii, jj, t0, t1, itrs, lm, ep, motion_only, EP, uniq=uniq)
        except RuntimeError:
            pass
        torch.cuda.synchronize()
        store["poses_out"] = v.poses.cpu().numpy(); store["disps_out"] = v.disps.cpu().numpy()
    graph._solve = solve
    orig_op = graph._operator
    def op(net, coords1, target_prev, feats, ii, corr0=None):
        store["corr0"] = corr0.float().cpu().numpy()[:, ::2, ::2, ::8]
        store["net_in"] = net.float().cpu().numpy()[:, ::2, ::2, ::8]
        store["coords1"] = coords1.cpu().numpy(); store["c_before"] = store["coords1"]; store["c_after_lookup"] = store["coords1"]
        r = orig_op(net, coords1, target_prev, feats, ii, corr0)
        torch.cuda.synchronize()
        store["net_out"] = net.float().cpu().numpy()
        store["dw"] = r[0].cpu().numpy()
        return r
    graph._operator = op


def worker(rank, world, port, out, corr, variant="dist"):
    import torch.distributed as dist
    g = T._graph("C2")
    if variant.startswith("dist"):
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from droid_amd.dist_graph import DistFactorGraph
        v, graph = T._setup(g, DistFactorGraph, "alt", False)
    else:                                            # the same shard in a plain FactorGraph, no process group, no BA
        from droid_amd.factor_graph import FactorGraph
        from droid_amd.dist_ba import shard_edges_by_source_frame
        sh, _ = shard_edges_by_source_frame(g["ii"], world)
        v, graph = T._setup(g, FactorGraph, "alt", False, order=sh[rank])
    graph.upsample = False
    st = {}
    instrument(graph, st)
    orig_rep = v.reproject
    def rep(ii, jj):
        a = orig_rep(ii, jj)
        st["_c_before"] = a[0].clone()                  # device copy on the same stream, right behind the kernel; no host sync
        st["_disps_before"] = v.disps.clone(); st["_poses_before"] = v.poses.clone()
        return a
    v.reproject = rep
    orig_pf = graph._pyramid_features
    def pf(block, coords1):
        r = orig_pf(block, coords1)
        st["_c_after_lookup"] = coords1.clone()
        return r
    graph._pyramid_features = pf
    if not variant.startswith("dist"):
        orig = graph._solve
        def solve(*a, **k):
            try:
                orig(*a, **k)
            except RuntimeError:
                pass
        graph._solve = solve
    if variant == "plain_barrier":                     # start the step in lock-step, like ranks leaving a rendezvous do
        import time
        torch.cuda.synchronize()
        t = float(open("/tmp/dfs_go").read())
        while time.time() < t:
            pass
    if variant == "dist_skew" and rank == 1:
        import time
        time.sleep(0.5)
    graph.update_lowmem(steps=1, corr=corr)
    torch.cuda.synchronize()
    cb, ca = st.pop("_c_before")[0], st.pop("_c_after_lookup")
    st["c_before"] = cb.cpu().numpy(); st["c_after_lookup"] = ca.cpu().numpy()
    db_, pb_ = st.pop("_disps_before"), st.pop("_poses_before")
    print("  [rank %d] coords right behind reproject vs after the lookup: %d elements differ; disps changed since: %s poses changed: %s" % (
        rank, int((cb != ca).sum()), bool((db_ != v.disps).any()) if variant.startswith("plain") else "n/a", "n/a"), flush=True)
    st["local_index"] = graph.local_index().cpu().numpy() if variant.startswith("dist") else sh[rank]
    np.savez(out % rank, **st)
    if variant.startswith("dist"):
        dist.destroy_process_group()


def main():
    from droid_amd.factor_graph import FactorGraph
    corr = sys.argv[1]
    variant = sys.argv[2] if len(sys.argv) > 2 else "dist"
    g = T._graph("C2")
    v, graph = T._setup(g, FactorGraph, "alt", False)
    graph.upsample = False
    ref = {}
    instrument(graph, ref)
    graph.update_lowmem(steps=1, corr=corr)
    for k in range(3):
        out = "/tmp/dfs_%d_%%d.npz" % k
        import time
        open("/tmp/dfs_go", "w").write(repr(time.time() + 12.0))
        mp.spawn(worker, args=(2, 29480 + k, out, corr, variant), nprocs=2, join=True)
        for r in (0, 1):
            s = np.load(out % r)
            ix = s["local_index"]
            for key in ("c_before", "c_after_lookup", "corr0", "net_in", "coords1", "net_out", "dw"):
                dd = np.abs(s[key] - ref[key][ix])
                print("   rank %d %s: max %.3g, elements > 2e-3: %d" % (r, key, dd.max(), int((dd > 2e-3).sum())), end="")
                if key in ("net_out", "dw") and (dd > 2e-3).any():
                    bad = np.argwhere(dd > 2e-3)
                    print("  first %s last %s; channels %s" % (bad[0].tolist(), bad[-1].tolist(), np.unique(bad[:, 3])[:20]), end="")
                print()
            dt = np.abs(s["tb"] - ref["tb"][ix])
            bad = np.argwhere(dt > 2e-3)
            print("   rank %d: %d elements of tb off by > 2e-3; edges %s; first %s" % (r, len(bad), np.unique(bad[:, 0])[:12], bad[:6].tolist()))
            print("rep %d rank %d: tb %.3g wb %.3g (vs single, same edges)  damping(own) %.3g  poses_in %.3g disps_in %.3g | poses_out %.3g disps_out %.3g" % (
                k, r, np.abs(s["tb"] - ref["tb"][ix]).max(), np.abs(s["wb"] - ref["wb"][ix]).max(),
                np.abs(s["damping"] - ref["damping"])[np.unique(s["ii"])].max(), np.abs(s["poses_in"] - ref["poses_in"]).max(),
                np.abs(s["disps_in"] - ref["disps_in"]).max(), np.abs(s["poses_out"] - ref["poses_out"]).max(), np.abs(s["disps_out"] - ref["disps_out"]).max()), flush=True)


if __name__ == "__main__":
    main()
