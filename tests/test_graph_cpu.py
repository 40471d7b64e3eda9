"""CPU: the oracle's factor-graph glue against the vectors written by the reference's own factor_graph.py
(tests/golden/graph_python.npz, tests/golden/make_graph_golden.py)."""
import os
import numpy as np

from oracle import graph as ograph, geom as ogeom
from golden_inputs import graph_scenario


def test_proximity_edges_match_reference_factor_graph(golden_dir):
    G = np.load(os.path.join(golden_dir, "graph_python.npz"))
    S = graph_scenario()
    t = S["n_frames"]
    es = ograph.proximity_edges(G["B_dist"], 0, 0, t, S["prox_rad"], S["prox_nms"], S["prox_thresh"], 16 * t, [])
    assert es == list(zip(G["B_ii"].tolist(), G["B_jj"].tolist()))


def test_frame_distance_matrix_matches_reference_orchestration(golden_dir):
    """DepthVideo.distance (bidirectional mean of two frame_distance calls, depth_video.py:181-211)"""
    G = np.load(os.path.join(golden_dir, "graph_python.npz"))
    S = graph_scenario()
    t = S["n_frames"]
    ii, jj = np.meshgrid(np.arange(t), np.arange(t), indexing="ij")
    ii = ii.reshape(-1); jj = jj.reshape(-1)
    d = 0.5 * (ogeom.frame_distance(S["poses"], S["disps"], S["intrinsics"][0], ii, jj, S["prox_beta"]) +
               ogeom.frame_distance(S["poses"], S["disps"], S["intrinsics"][0], jj, ii, S["prox_beta"]))
    assert np.abs(d - G["B_dist"]).max() <= 1e-5 * max(1.0, np.abs(G["B_dist"]).max())


def test_window_spread_separates_coherent_from_incoherent_flows():
    """host logic behind FactorGraph's layout fallback (droid_amd.corr.CorrBlock.window_spread): 1.0 for a translation, a little
    above for an affine / reprojection-like flow, far beyond CorrBlock.SPREAD_LIMIT for independent random coordinates"""
    import torch
    from droid_amd.corr import CorrBlock
    h, w = 48, 64
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    grid = torch.stack([xx, yy], -1)[None]
    assert CorrBlock.window_spread(grid + torch.tensor([3.3, -2.6])) == 1.0
    affine = grid * torch.tensor([1.05, 0.97]) + 0.03 * grid.flip(-1) + torch.tensor([-4.2, 1.7])
    assert 1.0 < CorrBlock.window_spread(affine) < 1.6
    gen = torch.Generator().manual_seed(0)
    rnd = torch.stack([torch.rand(3, h, w, generator=gen) * w, torch.rand(3, h, w, generator=gen) * h], -1)
    assert CorrBlock.window_spread(rnd) > 5 * CorrBlock.SPREAD_LIMIT
    # depth discontinuities: half of every 8x8 block displaced by 30 pixels -> (10 + 30) / 10 = 4x the window union in x
    step = grid.clone(); step[..., 0] += 30.0 * ((xx % 8) >= 4)
    assert 3.9 < CorrBlock.window_spread(step) < 4.1
    assert CorrBlock.window_spread(rnd[:0]) == 1.0


def test_committed_goldens_match_their_hash_manifest(golden_dir):
    """tests/golden/MANIFEST.sha256 pins every committed golden vector (outputs of the reference's own code, which only the build
    container can regenerate) to the bytes the tolerances were set against; `python tests/golden/manifest.py --write` after a
    deliberate regeneration"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("golden_manifest", os.path.join(golden_dir, "manifest.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    cur, want = m.current(), m.read()
    assert set(cur) == set(m.GENERATOR), sorted(set(m.GENERATOR) - set(cur))
    assert cur == want, [n for n in cur if cur[n] != want.get(n)]
