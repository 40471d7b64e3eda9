"""Oracle restatement of the factor-graph glue around the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  proximity_edges ... FactorGraph.add_proximity_factors' candidate selection   droid_slam/factor_graph.py:346-412
  motion_features ... cat(coords1 - coords0, target - coords1).clamp(-64, 64)   droid_slam/factor_graph.py:221-222
Pinned by tests/golden/graph_python.npz (the reference's own factor_graph.py run on CPU, tests/golden/make_graph_golden.py).
"""
import numpy as np


def proximity_edges(d, t0, t1, t, rad, nms, thresh, max_factors, existing, stereo=False):
    """d [(t-t0)*(t-t1)] frame distances (i in [t0,t) x j in [t1,t), row-major); existing = iterable of (i, j) edges already
    in the graph (active + bad + inactive).  Returns the list of (i, j) edges in the reference's order."""
    d = np.array(d, dtype=np.float64).copy()
    nj = t - t1
    ii, jj = np.meshgrid(np.arange(t0, t), np.arange(t1, t), indexing="ij")
    ii = ii.reshape(-1); jj = jj.reshape(-1)
    d[ii - rad < jj] = np.inf
    d[d > 100] = np.inf

    def suppress(i, j):
        for di in range(-nms, nms + 1):
            for dj in range(-nms, nms + 1):
                if abs(di) + abs(dj) <= max(min(abs(i - j) - 2, nms), 0):
                    i1, j1 = i + di, j + dj
                    if t0 <= i1 < t and t1 <= j1 < t:
                        d[(i1 - t0) * nj + (j1 - t1)] = np.inf
    for (i, j) in existing:
        suppress(int(i), int(j))
    es = []
    for i in range(t0, t):
        if stereo:
            es.append((i, i))
            d[(i - t0) * nj + (i - t1)] = np.inf
        for j in range(max(i - rad - 1, 0), i):
            es.append((i, j)); es.append((j, i))
            d[(i - t0) * nj + (j - t1)] = np.inf
    for k in np.argsort(d, kind="stable"):
        if d[k] > thresh:
            continue
        if max_factors > 0 and len(es) > max_factors:
            break
        i, j = int(ii[k]), int(jj[k])
        es.append((i, j)); es.append((j, i))
        suppress(i, j)
    return es


def motion_features(coords1, target):
    """coords1, target [E,h,w,2] -> [E,h,w,4]"""
    E, h, w, _ = coords1.shape
    y, x = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    c0 = np.stack([x, y], -1)[None]
    return np.clip(np.concatenate([coords1 - c0, target - coords1], -1), -64.0, 64.0)
