#!/usr/bin/env python
"""CPU model of the HBM lines the pyramid lookup reads (no GPU needed): for a sample of edges of a synthetic graph and a
flow (bench | planes | smooth_gt | random) it walks every source block (= wave, 64 pixels) and level, forms the set of 128-byte
lines the wave's tap loads touch under a given layout, and reports bytes per edge-pixel against the algorithmic 480 B (240 taps),
plus -- round 5 -- the number of maximal CONTIGUOUS runs those lines form per wave and level and their mean length (the
address path and the DRAM pages see runs, not lines).
Assumes what the kernel relies on: a line is fetched from HBM once per wave (L1/L2 absorb the re-touches inside a wave), no
reuse between waves except the shared all-zero row.  Validated against rocprofv3 counters (profiles/r02_lookup_pmc.json:
548 B/ep of reads on the bench flow; round 4's in-run pass: 6.98 GB / 12.58 M ep = 555 B/ep).

layouts (all 64 pixels per source block; `p` = pixel inside the block):
  pair       V'[sb 8x8][v][u/2][p][u&1]      the layout of csrc/corr_pyramid.hip (256-byte runs = one pair of cells x 64 pixels)
  cell       V'[sb 8x8][v][u][p]             one cell x 64 pixels = one 128-byte line
  pairodd    pair + an odd-aligned second copy: a lane takes the copy whose pairs start at its u0 (4 pairs always; 2x the memory)
  pair4x16   pair with 4-row x 16-column source blocks   (round 5, VERDICT r4 item 6 (i): the window union shrinks along y)
  pair16x4   pair with 16-row x 4-column source blocks   (                            ... along x)
  vpair      V'[sb 8x8][v/2][u/2][v&1][p][u&1]  two displacement rows interleaved: a window row PAIR is one 2.5-KB run (iii)
  pairmid    pair with the displacement origin in the MIDDLE of the cyclic row: u = (x2 - x1 + w2/2) mod w2.  Same lines; a small flow's
             window no longer straddles the wrap point u = 0, so a window row is one run instead of two
  l3whole    pair, but level 3 read whole: its slice per source block is 6 x 8 cells = 6 KB contiguous (ii); level 2's slice
             (12 x 16 cells = 24 KB per block = 384 B/ep against 128 B/ep of taps) would triple that level's bytes: not scored
usage: python scripts/lookup_traffic_model.py [--edges 48] [--config C3] [--layouts pair,cell,...] [--md]
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


BLOCK_SHAPE = {"pair4x16": (4, 16), "pair16x4": (16, 4)}


def lines_per_wave(coords, layout):
    """coords [n,2,h,w] -> (bytes read per edge-pixel, per-level list, runs per wave and level, mean run length in bytes)"""
    n, _, h, w = coords.shape
    per_level, runs_level, runlen_level = [], [], []
    y1, x1 = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    bh, bw = BLOCK_SHAPE.get(layout, (8, 8))
    sb = (y1 // bh) * (w // bw) + (x1 // bw)
    p = (y1 % bh) * bw + (x1 % bw)
    nblk = (h // bh) * (w // bw)
    base = "pair" if layout in ("pair4x16", "pair16x4", "l3whole", "pairmid") else layout
    for l in range(4):
        h2, w2 = h >> l, w >> l
        if layout == "l3whole" and l == 3:                                     # the whole slice of the block: one run
            nlines = h2 * (w2 // 2) * 2
            per_level.append(nlines * 128.0 * nblk / (h * w)); runs_level.append(1.0); runlen_level.append(nlines * 128.0)
            continue
        cx, cy = coords[:, 0] / 2 ** l, coords[:, 1] / 2 ** l
        X0 = np.clip(np.floor(cx), -65536, 65536).astype(np.int64) - 3
        Y0 = np.clip(np.floor(cy), -65536, 65536).astype(np.int64) - 3
        u0 = (X0 - (x1 >> l)[None] + (w2 // 2 if layout == "pairmid" else 0)) % w2
        keys = []
        for j in range(8):
            y2 = Y0 + j
            inside = (y2 >= 0) & (y2 < h2)
            v = np.where(inside, (y2 - (y1 >> l)[None]) % h2, h2)             # h2 = zero row (shared per edge-level: not counted)
            if base in ("pair", "vpair"):
                k0 = u0 >> 1
                for m in range(5):
                    mm = np.where(m == 4, 3 + (u0 & 1), m)
                    pair = (k0 + mm) % (w2 // 2)
                    if base == "pair":
                        key = ((v * (w2 // 2) + pair) * 2 + (p >> 5)[None])    # line address inside (edge, sb): the HBM order
                    else:
                        key = ((((v >> 1) * (w2 // 2) + pair) * 2 + (v & 1)) * 2 + (p >> 5)[None])
                    keys.append(np.where(inside, key, -1))
            elif base == "cell":
                for i in range(8):
                    u = (u0 + i) % w2
                    keys.append(np.where(inside, v * w2 + u, -1))
            elif base == "pairodd":
                par = u0 & 1
                k0 = (u0 - par) >> 1
                for m in range(4):
                    pair = (k0 + m) % (w2 // 2)
                    key = (((v * (w2 // 2) + pair) * 2 + (p >> 5)[None]) * 2 + par)
                    keys.append(np.where(inside, key, -1))
            else:
                raise ValueError(layout)
        K = np.stack(keys, 1)                                                  # [n, loads, h, w]
        # unique lines per (edge, source block); runs = maximal sequences of consecutive line addresses inside one
        grp = np.arange(n)[:, None, None, None] * nblk + sb[None, None]
        big = K.astype(np.int64) + (1 << 40) * grp
        big = np.unique(big[K >= 0])
        total = len(big)
        nruns = 1 + int(np.count_nonzero(np.diff(big) != 1)) if total else 0     # (a group change is a jump of ~2^40 as well)
        per_level.append(total * 128.0 / (n * h * w))
        runs_level.append(nruns / float(n * nblk))
        runlen_level.append(total * 128.0 / max(1, nruns))
    return sum(per_level), per_level, runs_level, runlen_level


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", type=int, default=48)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--layouts", default="pair,pairmid,cell,pairodd,pair4x16,pair16x4,vpair,l3whole")
    ap.add_argument("--md", action="store_true", help="markdown table (profiles/r05_lookup_layout_model.md)")
    a = ap.parse_args()
    g = syn.make_graph(a.config)
    F = flows(g, a.edges)
    layouts = a.layouts.split(",")
    if a.md:
        print("| flow | layout | B/ep read | x 480 | per level B/ep | runs per wave and level | mean run (bytes) per level |")
        print("|---|---|---|---|---|---|---|")
    else:
        print("bytes read per edge-pixel (algorithmic: 480 = 64+64+64+48 taps x 2 B); per level in brackets; runs per wave; mean run bytes")
    for name, c in F.items():
        for layout in layouts:
            tot, lv, rn, rl = lines_per_wave(c, layout)
            f = lambda xs, fmt: ", ".join(fmt % x for x in xs)
            if a.md:
                print("| %s | %s | %.1f | %.3f | %s | %s | %s |" % (name, layout, tot, tot / 480.0, f(lv, "%.1f"), f(rn, "%.1f"), f(rl, "%.0f")))
            else:
                print("%-10s %-8s %7.1f B/ep  x%.3f  [%s]  runs [%s]  run bytes [%s]" % (name, layout, tot, tot / 480.0, f(lv, "%.1f"), f(rn, "%.1f"), f(rl, "%.0f")))
