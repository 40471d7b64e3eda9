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
