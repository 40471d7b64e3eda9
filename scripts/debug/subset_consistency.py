"""diagnostic: one update_lowmem step on all C2 edges vs on a rank's subset, single process: per-edge hidden state / targets / weights of the
common edges must agree to ~1 fp16 ulp (no BA influence on them within the first step)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "droid-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_dist_graph_gpu as T
from droid_amd.factor_graph import FactorGraph
from droid_amd.dist_ba import shard_edges_by_source_frame
import droid_backends as db

mode = sys.argv[1] if len(sys.argv) > 1 else "pyramid"
g = T._graph("C2")
sh, b = shard_edges_by_source_frame(g["ii"], 2)


def run(order, fused=1):
    db.set_option("lookup_fused", fused)
    v, graph = T._setup(g, FactorGraph, "alt", False, order=order)
    graph.upsample = False
    graph._solve = lambda *a, **k: None          # (no BA: the first step's operator outputs do not depend on it)
    graph.update_lowmem(steps=1, corr=mode)
    torch.cuda.synchronize()
    return graph._net.float().cpu().numpy(), graph.target[0].cpu().numpy(), graph.weight[0].cpu().numpy()

full = run(None)
full2 = run(None)
print("full vs full: net %.3g target %.3g weight %.3g" % tuple(np.abs(a - b).max() for a, b in zip(full, full2)))
for fused in (1, 0):
    for r in (0, 1):
        for rep in range(2):
            sub = run(sh[r], fused)
            dn = np.abs(sub[0] - full[0][sh[r]]); dt = np.abs(sub[1] - full[1][sh[r]]); dw = np.abs(sub[2] - full[2][sh[r]])
            worst = np.argsort(dn.reshape(len(sh[r]), -1).max(1))[-3:]
            print("fused %d shard %d (%d edges) rep %d vs full: net max %.3g (edges>2^-8: %d) target max %.3g weight max %.3g; worst edges %s" % (
                fused, r, len(sh[r]), rep, dn.max(), int((dn.reshape(len(sh[r]), -1).max(1) > 2.0 ** -8).sum()), dt.max(), dw.max(), worst), flush=True)
