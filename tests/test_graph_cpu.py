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


def test_from_frames_prepares_each_frame_once_and_indexes_the_edges(monkeypatch):
    """host logic of droid_amd.corr.CorrBlock.from_frames (the kernels are replaced by recorders): only the frames that occur in the
    edge list are prepared, each once; edge e reads source frame ii[e] (camera 0) and target frame jj[e] -- camera 1 on a stereo
    self-edge (factor_graph.py:128-133: fmaps[jj, (ii == jj)]); canvases are zero-padded before the preparation; image sizes that are
    kept transposed or in strips take the per-edge constructor."""
    import torch
    from droid_amd import corr as corr_mod
    calls = {}

    class FakeBackend:
        @staticmethod
        def corr_pyramid_prepare_frames(f, h_real, w_real):
            calls["prep"] = (f.clone(), h_real, w_real)
            return f.reshape(f.shape[0], -1)                       # "prepared" = the frame itself, flattened

        @staticmethod
        def corr_pyramid_build_indexed(prep, i1, i2, h, w, out):
            calls["build"] = (i1.clone(), i2.clone(), h, w, out)
            return torch.stack([prep[i1], prep[i2]], 1)            # record e = (source frame, target frame)

        @staticmethod
        def corr_pyramid_build(f1, f2, *a):
            calls["per_edge"] = (f1.shape, f2.shape)
            return torch.zeros(f1.shape[0], 4)
    monkeypatch.setattr(corr_mod, "droid_backends", FakeBackend)
    N, rig, C, h, w = 7, 2, 128, 16, 32
    fmaps = torch.arange(N * rig, dtype=torch.float32).reshape(N, rig, 1, 1, 1).expand(N, rig, C, h, w).half()   # value = frame * rig + camera
    ii = torch.tensor([1, 1, 2, 4, 4, 5])
    jj = torch.tensor([2, 1, 1, 5, 4, 4])                         # two stereo self-edges: (1, 1), (4, 4)
    blk = corr_mod.CorrBlock.from_frames(fmaps, ii, jj)
    f, h_real, w_real = calls["prep"]
    assert (h_real, w_real) == (h, w) and f.shape == (6, C, h, w)                      # frames {1,2,4,5} x camera 0 + cameras 1 of frames 1 and 4
    assert sorted(f[:, 0, 0, 0].tolist()) == [2.0, 3.0, 4.0, 8.0, 9.0, 10.0]
    src, tgt = blk.pyramid[:, 0, 0], blk.pyramid[:, 1, 0]
    assert src.tolist() == [float(i * rig) for i in ii.tolist()]
    assert tgt.tolist() == [float(j * rig + (1 if i == j else 0)) for i, j in zip(ii.tolist(), jj.tolist())]
    assert calls["build"][2:4] == (16, 32) and (blk.hc, blk.wc, blk.transposed, blk.strips) == (16, 32, False, None)
    # mono buffer, image on a canvas: 30 x 40 -> 32 x 64, zero outside the image
    fm1 = torch.ones(3, 1, C, 30, 40).half()
    blk = corr_mod.CorrBlock.from_frames(fm1, torch.tensor([0, 1]), torch.tensor([1, 2]))
    f, h_real, w_real = calls["prep"]
    assert (h_real, w_real) == (30, 40) and f.shape == (3, C, 32, 64) and f[:, :, 30:].abs().sum() == 0 and f[:, :, :, 40:].abs().sum() == 0
    assert calls["build"][2:4] == (32, 64)
    # transposed / strip sizes: the per-edge constructor
    for hh, ww in ((41, 73), (72, 96)):
        calls.pop("per_edge", None)
        corr_mod.CorrBlock.from_frames(torch.ones(2, 1, C, hh, ww).half(), torch.tensor([0]), torch.tensor([1]))
        assert "per_edge" in calls
