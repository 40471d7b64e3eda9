#!/usr/bin/env python
"""CPU model of the HBM lines the pyramid lookup reads (no GPU needed): for a sample of edges of a synthetic graph and a
flow (bench | planes | random) it walks every 8x8 source block (= wave) and level, forms the set of 128-byte lines the
wave's tap loads touch under a given layout, and reports bytes per edge-pixel against the algorithmic 480 B (240 taps).
Assumes what the kernel relies on: a line is fetched from HBM once per wave (L1/L2 absorb the re-touches inside a wave), no
reuse between waves except the shared all-zero row.  Validated against rocprofv3 counters (profiles/r02_lookup_pmc.json:
548 B/ep of reads on the bench flow).

layouts:
  pair     V'[sb][v][u/2][p][u&1]     the layout of csrc/corr_pyramid.hip (256-byte runs = one pair of cells x 64 pixels)
  cell     V'[sb][v][u][p]            one cell x 64 pixels = one 128-byte line
  pairodd  pair + an odd-aligned second copy: a lane takes the copy whose pairs start at its u0 (4 pairs always)
usage: python scripts/lookup_traffic_model.py [--edges 48] [--config C3]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
import numpy as np
from droid_amd import synthetic as syn


def flows(g, n_edges, seed=11):
    rng = np.random.default_rng(seed)
    E = len(g["ii"])
    sel = np.sort(rng.choice(E, n_edges, replace=False))
    ii, jj = g["ii"][sel], g["jj"][sel]
    N, ht, wd = g["n_frames"], g["ht"], g["wd"]
    out = {}
    c, _ = syn.reproject(g["poses"].astype(np.float64), g["disps"].astype(np.float64), syn.INTRINSICS, ii, jj)
    out["bench"] = c.reshape(n_edges, 2, ht, wd)
    planes = np.kron(rng.uniform(0.3, 2.0, (N, ht // 16, wd // 16)), np.ones((16, 16)))
    c, _ = syn.reproject(g["poses_gt"].astype(np.float64), planes, syn.INTRINSICS, ii, jj)
    out["planes"] = c.reshape(n_edges, 2, ht, wd)
    c, _ = syn.reproject(g["poses_gt"].astype(np.float64), g["disps_gt"].astype(np.float64), syn.INTRINSICS, ii, jj)
    out["smooth_gt"] = c.reshape(n_edges, 2, ht, wd)
    out["random"] = np.stack([rng.uniform(0, wd, (n_edges, ht, wd)), rng.uniform(0, ht, (n_edges, ht, wd))], 1)
    return out


def lines_per_wave(coords, layout):
    """coords [n,2,h,w] -> (bytes read per edge-pixel, per-level list) under `layout`"""
    n, _, h, w = coords.shape
    per_level = []
    y1, x1 = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    sb = (y1 // 8) * (w // 8) + (x1 // 8)
    p = (y1 % 8) * 8 + (x1 % 8)
    nblk = (h // 8) * (w // 8)
    for l in range(4):
        h2, w2 = h >> l, w >> l
        cx, cy = coords[:, 0] / 2 ** l, coords[:, 1] / 2 ** l
        X0 = np.clip(np.floor(cx), -65536, 65536).astype(np.int64) - 3
        Y0 = np.clip(np.floor(cy), -65536, 65536).astype(np.int64) - 3
        u0 = (X0 - (x1 >> l)[None]) % w2
        total = 0
        keys = []
        for j in range(8):
            y2 = Y0 + j
            inside = (y2 >= 0) & (y2 < h2)
            v = np.where(inside, (y2 - (y1 >> l)[None]) % h2, h2)             # h2 = zero row (shared per edge-level: not counted)
            if layout == "pair":
                k0 = u0 >> 1
                for m in range(5):
                    mm = np.where(m == 4, 3 + (u0 & 1), m)
                    pair = (k0 + mm) % (w2 // 2)
                    key = ((v * (w2 // 2) + pair) * 2 + (p >> 5)[None])        # line id inside (edge, sb)
                    keys.append(np.where(inside, key, -1))
            elif layout == "cell":
                for i in range(8):
                    u = (u0 + i) % w2
                    keys.append(np.where(inside, v * w2 + u, -1))
            elif layout == "pairodd":
                par = u0 & 1
                k0 = (u0 - par) >> 1
                for m in range(4):
                    pair = (k0 + m) % (w2 // 2)
                    key = (((v * (w2 // 2) + pair) * 2 + (p >> 5)[None]) * 2 + par)
                    keys.append(np.where(inside, key, -1))
            else:
                raise ValueError(layout)
        K = np.stack(keys, 1)                                                  # [n, loads, h, w]
        # unique lines per (edge, source block)
        big = K.astype(np.int64) + (1 << 40) * (np.arange(n)[:, None, None, None] * nblk + sb[None, None])
        big = big[K >= 0]
        total = len(np.unique(big))
        per_level.append(total * 128.0 / (n * h * w))
    return sum(per_level), per_level


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", type=int, default=48)
    ap.add_argument("--config", default="C3")
    a = ap.parse_args()
    g = syn.make_graph(a.config)
    F = flows(g, a.edges)
    print("bytes read per edge-pixel (algorithmic: 480 = 64+64+64+48 taps x 2 B); per level in brackets")
    for name, c in F.items():
        for layout in ("pair", "cell", "pairodd"):
            tot, lv = lines_per_wave(c, layout)
            print("%-10s %-8s %7.1f B/ep  x%.3f  [%s]" % (name, layout, tot, tot / 480.0, ", ".join("%.1f" % x for x in lv)))
