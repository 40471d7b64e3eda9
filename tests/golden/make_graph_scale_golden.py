#!/usr/bin/env python
"""Generate tests/golden/graph_c2_python.npz and graph_stereo_python.npz by running the REFERENCE's own factor_graph.py +
depth_video.py + modules/corr.py + droid_net.py, unmodified, on CPU (build container only: needs /root/reference):

    python tests/golden/make_graph_scale_golden.py [stereo] [c2] [tum] [wide] [big] [c3] [c5] [--probe]

--probe: the same scenario with the stored feature / hidden-state / context maps (fmaps, nets, inps) moved by ONE
fp16 ulp on half of their values (seeded) -> /tmp/graph_<cfg>_probe.npz, and the movement of every quantity the composed tests
of tests/test_scale_gpu.py assert, in the tests' own metrics, -> tests/golden/graph_scale_probe.json.  That file calibrates the
tests' tolerances: an implementation that rounds differently somewhere (accumulation order, fp16 stores) moves the outputs by
about what a rounding-level change of the inputs does.

Same replacements as make_graph_golden.py (droid_backends -> oracle-backed shim, lietorch / torch_scatter shims, the two
hard-coded "cuda" devices); the update operator runs under torch.autocast(fp16) like under factor_graph.py's decorators.

Scenario C2 (BASELINE configs[1] at FULL size: 64 keyframes / 512 edges / 48x64, the seeded synthetic graph of
droid_amd.synthetic.make_graph("C2", with_features=True)): FactorGraph.add_factors on all 512 edges, then two
FactorGraph.update iterations (factor_graph.py:214-263).  Kept per iteration: poses, depths, damping of every frame; the
hidden state / target / weight of 12 sample edges (first, middle, last four) and per-edge means of |target - coords0| and
weight for ALL edges (the full hidden state would be 200 MB).
Scenario S (stereo, BASELINE configs[4]'s ingredients at 6 keyframes / 16x64): DepthVideo(stereo=True), fmaps of both
cameras, stereo self-edges (i, i) + temporal edges, two update iterations: pins `fmaps[jj, c]` (factor_graph.py:128-133) and
the stereo branch of the BA (droid_kernels.cu:228-238) inside one composed iteration.
"""
import os
import sys
import time
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_graph_golden as base                   # sets up sys.path, the shims and the two device patches

from droid_amd import synthetic as syn
from droid_amd.weights import fill_deterministic
from golden_inputs import graph_scenario, C2_SAMPLE_EDGES, stereo_scenario

PROBE = "--probe" in sys.argv
ref_dv, ref_fg, ref_net = base.ref_dv, base.ref_fg, base.ref_net


def perturb_one_ulp(g, seed=77):
    """fmaps / nets / inps (fp16) moved by one ulp on half of their values: the bit pattern +-1 (sign-magnitude, so +1 on the
    pattern = one ulp away from zero, -1 = one ulp towards zero); zeros and the largest finite values are left alone"""
    rng = np.random.default_rng(seed)
    for k in ("fmaps", "nets", "inps"):
        a = np.ascontiguousarray(g[k]).astype(np.float16)
        bits = a.view(np.uint16).copy()
        mag = bits & 0x7FFF
        move = rng.random(bits.shape) < 0.5
        up = rng.random(bits.shape) < 0.5
        ok = move & (mag > 0) & (mag < 0x7BFF)
        bits = np.where(ok & up, bits + 1, np.where(ok & ~up, bits - 1, bits)).astype(np.uint16)
        g[k] = bits.view(np.float16).reshape(a.shape)
    return g


def _rot_angle(q, qr):
    v = q[:, 3:4] * -qr[:, :3] + qr[:, 3:4] * q[:, :3] + np.cross(q[:, :3], -qr[:, :3])
    return 2 * np.linalg.norm(v, axis=-1)


def probe_report(cfg, G, P):
    """movement golden -> perturbed run, per iteration, in the metrics of tests/test_scale_gpu.py's composed tests"""
    import json
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    rep = {}
    for tag in ("U1", "U2"):
        r = {}
        p, rp = f64(P[tag + "_poses"]), f64(G[tag + "_poses"])
        r["pose_trans_max"] = float(np.abs(p[:, :3] - rp[:, :3]).max())
        r["pose_rot_max_rad"] = float(_rot_angle(p[:, 3:], rp[:, 3:]).max())
        dk = tag + ("_disps" if tag + "_disps" in G else "_disps_f")
        e = np.abs(f64(P[dk]) - f64(G[dk])) / np.maximum(1.0, np.abs(f64(G[dk])))
        r["disps_rel_q99"], r["disps_rel_max"] = float(np.quantile(e, 0.99)), float(e.max())
        if tag + "_disps_mean" in G:
            r["disps_frame_mean_max"] = float(np.abs(f64(P[tag + "_disps_mean"]) - f64(G[tag + "_disps_mean"])).max())
        r["net_s_max"] = float(np.abs(f64(P[tag + "_net_s"]) - f64(G[tag + "_net_s"])).max())
        t = np.abs(f64(P[tag + "_target_s"]) - f64(G[tag + "_target_s"]))
        r["target_s_q999"], r["target_s_max"] = float(np.quantile(t, 0.999)), float(t.max())
        r["weight_s_max"] = float(np.abs(f64(P[tag + "_weight_s"]) - f64(G[tag + "_weight_s"])).max())
        mk = tag + ("_damping" if tag + "_damping" in G else "_damping_f")
        r["damping_rel_max"] = float(np.abs(f64(P[mk]) - f64(G[mk])).max() / np.abs(f64(G[mk])).max())
        if tag + "_damping_mean" in G:
            r["damping_frame_mean_rel_max"] = float(np.abs(f64(P[tag + "_damping_mean"]) - f64(G[tag + "_damping_mean"])).max() / np.abs(f64(G[tag + "_damping_mean"])).max())
        r["flow_mean_max"] = float(np.abs(f64(P[tag + "_flow_mean"]) - f64(G[tag + "_flow_mean"])).max())
        r["weight_mean_max"] = float(np.abs(f64(P[tag + "_weight_mean"]) - f64(G[tag + "_weight_mean"])).max())
        r["net_absmean_max"] = float(np.abs(f64(P[tag + "_net_absmean"]) - f64(G[tag + "_net_absmean"])).max())
        rep[tag] = r
        print("probe %s %s: %s" % (cfg, tag, "  ".join("%s %.2e" % kv for kv in r.items())), flush=True)
    path = os.path.join(HERE, "graph_scale_probe.json")
    allp = json.load(open(path)) if os.path.exists(path) else {}
    allp[cfg] = rep
    allp["_about"] = ("movement of the composed goldens (reference factor_graph.py on CPU, tests/golden/make_graph_scale_golden.py) when "
                      "fmaps / nets / inps move by one fp16 ulp on half of their values (--probe); metrics = those asserted by "
                      "tests/test_scale_gpu.py::test_composed_update_at_c{2,3}_matches_reference_factor_graph")
    json.dump(allp, open(path, "w"), indent=1, sort_keys=True)
def small_metrics(A, B, tag):
    """deviation of run A from run B of one of the small scenarios (stereo / tum / wide / big), in the metrics the composed tests of
    tests/test_scale_gpu.py assert (`_composed_small_metrics` there computes the same from the product's tensors)"""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    r = {}
    p, rp = f64(A[tag + "_poses"]), f64(B[tag + "_poses"])
    r["pose_trans_max"] = float(np.abs(p[:, :3] - rp[:, :3]).max())
    r["pose_rot_max_rad"] = float(_rot_angle(p[:, 3:], rp[:, 3:]).max())
    e = np.abs(f64(A[tag + "_disps"]) - f64(B[tag + "_disps"])) / np.maximum(1.0, np.abs(f64(B[tag + "_disps"])))
    r["disps_rel_q99"], r["disps_rel_max"] = float(np.quantile(e, 0.99)), float(e.max())
    sfx = "" if tag + "_net" in B else "_s"
    r["net_s_max"] = float(np.abs(f64(A[tag + "_net" + sfx]) - f64(B[tag + "_net" + sfx])).max())
    t = np.abs(f64(A[tag + "_target" + sfx]) - f64(B[tag + "_target" + sfx]))
    r["target_s_q999"], r["target_s_max"] = float(np.quantile(t, 0.999)), float(t.max())
    r["weight_s_max"] = float(np.abs(f64(A[tag + "_weight" + sfx]) - f64(B[tag + "_weight" + sfx])).max())
    r["damping_rel_max"] = float(np.abs(f64(A[tag + "_damping"]) - f64(B[tag + "_damping"])).max() / np.abs(f64(B[tag + "_damping"])).max())
    if tag + "_flow_mean" in B:
        r["flow_mean_max"] = float(np.abs(f64(A[tag + "_flow_mean"]) - f64(B[tag + "_flow_mean"])).max())
        r["weight_mean_max"] = float(np.abs(f64(A[tag + "_weight_mean"]) - f64(B[tag + "_weight_mean"])).max())
    e = np.abs(f64(A[tag + "_disps_up"]) - f64(B[tag + "_disps_up"])) / np.maximum(1.0, np.abs(f64(B[tag + "_disps_up"])))
    r["disps_up_rel_q99"], r["disps_up_rel_max"] = float(np.quantile(e, 0.99)), float(e.max())
    return r


def probe_report_small(name, fname, P):
    """--probe for a small scenario: movement golden -> run with one-ulp-perturbed features, into graph_scale_probe.json[name]"""
    import json
    G = np.load(os.path.join(HERE, fname))
    rep = {}
    for tag in ("U1", "U2"):
        rep[tag] = small_metrics(P, G, tag)
        print("probe %s %s: %s" % (name, tag, "  ".join("%s %.2e" % kv for kv in rep[tag].items())), flush=True)
    path = os.path.join(HERE, "graph_scale_probe.json")
    allp = json.load(open(path)) if os.path.exists(path) else {}
    allp[name] = rep
    json.dump(allp, open(path, "w"), indent=1, sort_keys=True)


import modules.corr as ref_corr                    # reference (already imported by factor_graph.py)


def update_operator(seed):
    m = ref_net.UpdateModule()
    fill_deterministic(m, seed=seed)
    m.agg.eta[2] = base._SoftplusF32()
    m.eval()

    def update_op(*a, **kw):
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
            return m(*a, **kw)
    return update_op


def snap(out, tag, video, fg, N, sample=None):
    out[tag + "_poses"] = video.poses[:N].numpy().copy(); out[tag + "_disps"] = video.disps[:N].numpy().copy()
    out[tag + "_damping"] = fg.damping[:N].numpy().copy()
    tgt, wgt, net = fg.target[0], fg.weight[0], fg.net[0]
    if sample is None:
        out[tag + "_target"] = tgt.numpy().copy(); out[tag + "_weight"] = wgt.numpy().copy()
        out[tag + "_net"] = net.float().numpy().astype(np.float16)
    else:
        out[tag + "_target_s"] = tgt[sample].numpy().copy(); out[tag + "_weight_s"] = wgt[sample].numpy().copy()
        out[tag + "_net_s"] = net[sample].float().numpy().astype(np.float16)
        out[tag + "_flow_mean"] = (tgt - fg.coords0).abs().mean(dim=(1, 2, 3)).numpy().copy()
        out[tag + "_weight_mean"] = wgt.mean(dim=(1, 2, 3)).numpy().copy()
        out[tag + "_net_absmean"] = net.float().abs().mean(dim=(1, 2, 3)).numpy().copy()


def scenario_c2():
    g = syn.make_graph("C2", with_features=True)
    if PROBE:
        g = perturb_one_ulp(g)
    N, ht, wd = g["n_frames"], g["ht"], g["wd"]
    video = ref_dv.DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, stereo=False, device="cpu")
    video.poses[:N] = torch.as_tensor(g["poses"]); video.disps[:N] = torch.as_tensor(g["disps"])
    video.intrinsics[:N] = torch.as_tensor(g["intrinsics"])
    video.fmaps[:N] = torch.as_tensor(g["fmaps"]); video.nets[:N] = torch.as_tensor(g["nets"]); video.inps[:N] = torch.as_tensor(g["inps"])
    video.counter.value = N
    out = {}
    with torch.no_grad():
        fg = ref_fg.FactorGraph(video, update_operator(1234), device="cpu", corr_impl="volume", max_factors=-1, upsample=False)
        t = time.time()
        fg.add_factors(torch.as_tensor(g["ii"]), torch.as_tensor(g["jj"]))
        print("C2 add_factors (512 all-pairs volumes): %.1f s" % (time.time() - t), flush=True)
        assert len(fg.ii) == len(g["ii"])
        out["ii"], out["jj"] = fg.ii.numpy().copy(), fg.jj.numpy().copy()
        sample = torch.as_tensor(C2_SAMPLE_EDGES)
        for k in (1, 2):
            t = time.time()
            fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
            print("C2 update %d: %.1f s" % (k, time.time() - t), flush=True)
            snap(out, "U%d" % k, video, fg, N, sample)
    if PROBE:
        np.savez_compressed("/tmp/graph_c2_probe.npz", **out)
        return probe_report("C2", np.load(os.path.join(HERE, "graph_c2_python.npz")), out)
    np.savez_compressed(os.path.join(HERE, "graph_c2_python.npz"), **out)
    print("graph_c2: |dpose| %.3e %.3e  |ddisp| %.3e %.3e" % (
        np.abs(out["U1_poses"] - g["poses"]).max(), np.abs(out["U2_poses"] - g["poses"]).max(),
        np.abs(out["U1_disps"] - g["disps"]).max(), np.abs(out["U2_disps"] - g["disps"]).max()))


def scenario_stereo():
    S = stereo_scenario()
    if PROBE:
        S = perturb_one_ulp(S)
    N, ht, wd = S["n_frames"], S["ht"], S["wd"]
    video = ref_dv.DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, stereo=True, device="cpu")
    video.poses[:N] = torch.as_tensor(S["poses"]); video.disps[:N] = torch.as_tensor(S["disps"])
    video.intrinsics[:N] = torch.as_tensor(S["intrinsics"])
    video.fmaps[:N] = torch.as_tensor(S["fmaps"]); video.nets[:N] = torch.as_tensor(S["nets"]); video.inps[:N] = torch.as_tensor(S["inps"])
    video.counter.value = N
    out = {}
    with torch.no_grad():
        fg = ref_fg.FactorGraph(video, update_operator(S["weight_seed"]), device="cpu", corr_impl="volume", max_factors=-1, upsample=True)
        fg.add_factors(torch.as_tensor(S["ii"]), torch.as_tensor(S["jj"]))
        out["ii"], out["jj"] = fg.ii.numpy().copy(), fg.jj.numpy().copy()
        out["target0"] = fg.target[0].numpy().copy()
        for k in (1, 2):
            fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
            snap(out, "U%d" % k, video, fg, N)
            out["U%d_disps_up" % k] = video.disps_up[:N].numpy().astype(np.float16)
    if PROBE:
        return probe_report_small("stereo", "graph_stereo_python.npz", out)
    np.savez_compressed(os.path.join(HERE, "graph_stereo_python.npz"), **out)
    print("graph_stereo: %d edges (%d stereo); |dpose| %.3e %.3e" % (
        len(out["ii"]), int((out["ii"] == out["jj"]).sum()), np.abs(out["U1_poses"] - S["poses"]).max(), np.abs(out["U2_poses"] - S["poses"]).max()))


def scenario_tum_size():
    """scenario N: 6 keyframes at 30 x 40 (TUM's 240 x 320 images): neither the pyramid layout (w in {16,32,64}, h % 8 == 0) nor
    the production convolution tiling (w == 64) applies -> reference-layout volumes with FLOOR pooling (30x40, 15x20, 7x10,
    3x5), the generic convolution loop, per-edge context features.  Two update iterations with upsampling."""
    S = graph_scenario(6, 30, 40)
    if PROBE:
        S = perturb_one_ulp(S)
    N, ht, wd = S["n_frames"], S["ht"], S["wd"]
    video = ref_dv.DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, stereo=False, device="cpu")
    video.poses[:N] = torch.as_tensor(S["poses"]); video.disps[:N] = torch.as_tensor(S["disps"])
    video.intrinsics[:N] = torch.as_tensor(S["intrinsics"])
    video.fmaps[:N, 0] = torch.as_tensor(S["fmaps"]); video.nets[:N] = torch.as_tensor(S["nets"]); video.inps[:N] = torch.as_tensor(S["inps"])
    video.counter.value = N
    out = {}
    with torch.no_grad():
        fg = ref_fg.FactorGraph(video, update_operator(S["weight_seed"]), device="cpu", corr_impl="volume", max_factors=-1, upsample=True)
        fg.add_neighborhood_factors(0, N, r=2)
        out["ii"], out["jj"] = fg.ii.numpy().copy(), fg.jj.numpy().copy()
        out["target0"] = fg.target[0].numpy().copy()
        for k in (1, 2):
            fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
            snap(out, "U%d" % k, video, fg, N)
            out["U%d_disps_up" % k] = video.disps_up[:N].numpy().astype(np.float16)
    if PROBE:
        return probe_report_small("tum", "graph_tum_size_python.npz", out)
    np.savez_compressed(os.path.join(HERE, "graph_tum_size_python.npz"), **out)
    print("graph_tum_size: %d edges; |dpose| %.3e %.3e" % (len(out["ii"]), np.abs(out["U1_poses"] - S["poses"]).max(), np.abs(out["U2_poses"] - S["poses"]).max()))


WIDE_SAMPLE_EDGES = [0, 5, 13]


def scenario_big():
    """scenario G: 4 keyframes at 72 x 96 (a 576 x 768 input): more than 64 columns AND rows -> the HIP path keeps the pyramid in
    64-column strips (CorrBlock.strips) and runs the update operator through its generic loop.  Same recipe as scenario W."""
    _scenario_sampled(4, 72, 96, "graph_big_python.npz", "graph_big", [0, 3, 9])


def _scenario_sampled(n_frames, ht, wd, fname, label, sample_edges):
    S = graph_scenario(n_frames, ht, wd)
    if PROBE:
        S = perturb_one_ulp(S)
    N = S["n_frames"]
    video = ref_dv.DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, stereo=False, device="cpu")
    video.poses[:N] = torch.as_tensor(S["poses"]); video.disps[:N] = torch.as_tensor(S["disps"])
    video.intrinsics[:N] = torch.as_tensor(S["intrinsics"])
    video.fmaps[:N, 0] = torch.as_tensor(S["fmaps"]); video.nets[:N] = torch.as_tensor(S["nets"]); video.inps[:N] = torch.as_tensor(S["inps"])
    video.counter.value = N
    out = {}
    with torch.no_grad():
        fg = ref_fg.FactorGraph(video, update_operator(S["weight_seed"]), device="cpu", corr_impl="volume", max_factors=-1, upsample=True)
        fg.add_neighborhood_factors(0, N, r=2)
        out["ii"], out["jj"] = fg.ii.numpy().copy(), fg.jj.numpy().copy()
        out["target0_s"] = fg.target[0][sample_edges].numpy().copy()
        for k in (1, 2):
            fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
            snap(out, "U%d" % k, video, fg, N, sample=sample_edges)
            out["U%d_net_s" % k] = out["U%d_net_s" % k][:, :, ::3, ::3].copy()              # (every third pixel: 0.6 instead of 5.3 MB)
            out["U%d_disps_up" % k] = video.disps_up[:N].numpy()[:, ::4, ::4].astype(np.float16)
    if PROBE:
        return probe_report_small(label.replace("graph_", ""), fname, out)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print("%s: %d edges; |dpose| %.3e %.3e" % (label, len(out["ii"]), np.abs(out["U1_poses"] - S["poses"]).max(), np.abs(out["U2_poses"] - S["poses"]).max()))


def scenario_wide():
    """scenario W: 5 keyframes at 41 x 73 (a 16:9 video at the reference's demo resolution, demo.py:33-40: 1080p -> 328 x 584):
    more than 64 columns, so the HIP path keeps the image TRANSPOSED on its 64-column canvases (CorrBlock.transposed,
    UpdateModule.transposed_twin).  Two update iterations with upsampling; hidden state / target / weight of three edges and
    per-edge means of all edges."""
    S = graph_scenario(5, 41, 73)
    if PROBE:
        S = perturb_one_ulp(S)
    N, ht, wd = S["n_frames"], S["ht"], S["wd"]
    video = ref_dv.DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, stereo=False, device="cpu")
    video.poses[:N] = torch.as_tensor(S["poses"]); video.disps[:N] = torch.as_tensor(S["disps"])
    video.intrinsics[:N] = torch.as_tensor(S["intrinsics"])
    video.fmaps[:N, 0] = torch.as_tensor(S["fmaps"]); video.nets[:N] = torch.as_tensor(S["nets"]); video.inps[:N] = torch.as_tensor(S["inps"])
    video.counter.value = N
    out = {}
    with torch.no_grad():
        fg = ref_fg.FactorGraph(video, update_operator(S["weight_seed"]), device="cpu", corr_impl="volume", max_factors=-1, upsample=True)
        fg.add_neighborhood_factors(0, N, r=2)
        out["ii"], out["jj"] = fg.ii.numpy().copy(), fg.jj.numpy().copy()
        out["target0_s"] = fg.target[0][WIDE_SAMPLE_EDGES].numpy().copy()
        for k in (1, 2):
            fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
            snap(out, "U%d" % k, video, fg, N, sample=WIDE_SAMPLE_EDGES)
            out["U%d_disps_up" % k] = video.disps_up[:N].numpy().astype(np.float16)
    if PROBE:
        return probe_report_small("wide", "graph_wide_python.npz", out)
    np.savez_compressed(os.path.join(HERE, "graph_wide_python.npz"), **out)
    print("graph_wide: %d edges; |dpose| %.3e %.3e" % (len(out["ii"]), np.abs(out["U1_poses"] - S["poses"]).max(), np.abs(out["U2_poses"] - S["poses"]).max()))


# --------------------------------------------------------------------------------------------- C3 (the headline size)
class _ChunkedCorrBlock:
    """The reference's OWN modules/corr.py CorrBlock, instantiated per 64-edge chunk inside __call__ instead of once for all
    edges: 4096 all-pairs pyramids are 103 GB in fp16, the build container has 62 GB of RAM.  A correlation volume and its
    lookup are per-edge independent (corr.py:23-50), so the concatenated result is what CorrBlock(fmap1, fmap2)(coords)
    returns.  Only the constructor name `CorrBlock` in factor_graph.py's namespace is rebound; factor_graph.py itself runs
    unmodified."""
    CHUNK = 64

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3):
        self.fmap1, self.fmap2, self.num_levels, self.radius = fmap1, fmap2, num_levels, radius

    def __call__(self, coords):
        outs = []
        for s in range(0, self.fmap1.shape[1], self.CHUNK):
            blk = ref_corr.CorrBlock(self.fmap1[:, s:s + self.CHUNK], self.fmap2[:, s:s + self.CHUNK], self.num_levels, self.radius)
            outs.append(blk(coords[:, s:s + self.CHUNK]))
            del blk
        return torch.cat(outs, dim=1)

    def cat(self, other):
        self.fmap1 = torch.cat([self.fmap1, other.fmap1], 1); self.fmap2 = torch.cat([self.fmap2, other.fmap2], 1)
        return self

    def __getitem__(self, index):
        self.fmap1, self.fmap2 = self.fmap1[:, index], self.fmap2[:, index]
        return self


def update_operator_by_frame_groups(seed, frames_per_group=32):
    """The reference's UpdateModule (droid_net.py:78-143) under fp16 autocast, evaluated on groups of WHOLE source frames:
    its convolutions are per-edge, GraphAgg's scatter_mean (droid_net.py:62-67) is per source frame and returns rows in
    sorted-unique order, so groups of ascending frame ranges concatenate to the full-batch result; the full batch of 4096
    edges needs ~60 GB of fp16 activations on the host."""
    m = ref_net.UpdateModule()
    fill_deterministic(m, seed=seed)
    m.agg.eta[2] = base._SoftplusF32()
    m.eval()

    def update_op(net, inp, corr, flow, ii, jj):
        frames = torch.unique(ii)
        E = ii.shape[0]
        net_o = torch.empty_like(net)
        delta_o = weight_o = None
        etas, ups = [], []
        for s in range(0, len(frames), frames_per_group):
            grp = frames[s:s + frames_per_group]
            sel = torch.nonzero((ii >= grp[0]) & (ii <= grp[-1]))[:, 0]
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
                n, d, w, eta, up = m(net[:, sel], inp[:, sel], corr[:, sel], flow[:, sel], ii[sel], jj[sel])
            if delta_o is None:
                delta_o = torch.empty((1, E) + tuple(d.shape[2:]), dtype=d.dtype)
                weight_o = torch.empty((1, E) + tuple(w.shape[2:]), dtype=w.dtype)
            net_o[:, sel] = n.to(net_o.dtype); delta_o[:, sel] = d; weight_o[:, sel] = w
            etas.append(eta); ups.append(up)
        return net_o, delta_o, weight_o, torch.cat(etas, 1), torch.cat(ups, 1)
    return update_op


def scenario_c3():
    """BASELINE configs[2] at FULL size (the configuration bench.py times): 512 keyframes / 4096 edges / 48x64,
    droid_amd.synthetic.make_graph("C3", with_features=True); FactorGraph.add_factors on all edges, then two
    FactorGraph.update iterations (factor_graph.py:214-263).  Kept per iteration: poses and damping of every frame, depths
    of every 8th frame + per-frame depth means of all frames; target / weight of C3_SAMPLE_EDGES (64 edges spread over the
    graph) and their hidden state at every 4th pixel in x and y; per-edge means of |target - coords0|, weight and
    |hidden state| for ALL 4096 edges."""
    from golden_inputs import C3_SAMPLE_EDGES, C3_SAMPLE_FRAMES
    g = syn.make_graph("C3", with_features=True)
    if PROBE:
        g = perturb_one_ulp(g)
    N, ht, wd = g["n_frames"], g["ht"], g["wd"]
    video = ref_dv.DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, stereo=False, device="cpu")
    video.poses[:N] = torch.as_tensor(g["poses"]); video.disps[:N] = torch.as_tensor(g["disps"])
    video.intrinsics[:N] = torch.as_tensor(g["intrinsics"])
    video.fmaps[:N] = torch.as_tensor(g["fmaps"]); video.nets[:N] = torch.as_tensor(g["nets"]); video.inps[:N] = torch.as_tensor(g["inps"])
    video.counter.value = N
    out = {}
    ref_fg.CorrBlock = _ChunkedCorrBlock
    try:
        with torch.no_grad():
            fg = ref_fg.FactorGraph(video, update_operator_by_frame_groups(1234), device="cpu", corr_impl="volume", max_factors=-1, upsample=False)
            fg.add_factors(torch.as_tensor(g["ii"]), torch.as_tensor(g["jj"]))
            assert len(fg.ii) == len(g["ii"]) == 4096
            out["ii"], out["jj"] = fg.ii.numpy().copy(), fg.jj.numpy().copy()
            sample = torch.as_tensor(C3_SAMPLE_EDGES); fr = np.asarray(C3_SAMPLE_FRAMES)
            for k in (1, 2):
                t = time.time()
                fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
                print("C3 update %d: %.1f s" % (k, time.time() - t), flush=True)
                tag = "U%d" % k
                out[tag + "_poses"] = video.poses[:N].numpy().copy()
                d = video.disps[:N].numpy()
                out[tag + "_disps_f"] = d[fr].copy(); out[tag + "_disps_mean"] = d.reshape(N, -1).mean(1).copy()
                out[tag + "_damping_f"] = fg.damping[:N].numpy()[fr].copy()
                out[tag + "_damping_mean"] = fg.damping[:N].numpy().reshape(N, -1).mean(1).copy()
                tgt, wgt, net = fg.target[0], fg.weight[0], fg.net[0]
                out[tag + "_target_s"] = tgt[sample].numpy().copy(); out[tag + "_weight_s"] = wgt[sample].numpy().astype(np.float16)
                out[tag + "_net_s"] = net[sample][:, :, ::4, ::4].float().numpy().astype(np.float16)
                out[tag + "_flow_mean"] = (tgt - fg.coords0).abs().mean(dim=(1, 2, 3)).numpy().copy()
                out[tag + "_weight_mean"] = wgt.mean(dim=(1, 2, 3)).numpy().copy()
                out[tag + "_net_absmean"] = torch.stack([net[s:s + 256].float().abs().mean(dim=(1, 2, 3)) for s in range(0, net.shape[0], 256)]).reshape(-1).numpy().copy()
    finally:
        ref_fg.CorrBlock = ref_corr.CorrBlock
    if PROBE:
        np.savez_compressed("/tmp/graph_c3_probe.npz", **out)
        return probe_report("C3", np.load(os.path.join(HERE, "graph_c3_python.npz")), out)
    np.savez_compressed(os.path.join(HERE, "graph_c3_python.npz"), **out)
    print("graph_c3: |dpose| %.3e %.3e" % (np.abs(out["U1_poses"] - g["poses"]).max(), np.abs(out["U2_poses"] - g["poses"]).max()))


# --------------------------------------------------------------------------------------------- C5 (BASELINE configs[4])
def scenario_c5():
    """BASELINE configs[4] at FULL size: 1024 keyframes / 8192 edges / 48x64, STEREO + sensor depth + per-pixel depth-confidence weights
    (droid_amd.synthetic.make_graph("C5", with_features=True): `disps_conf`, seeded, non-constant), through the reference's GLOBAL-BA
    iteration -- FactorGraph.update_lowmem (factor_graph.py:266-330: alt-correlation in chunks of 8 source frames, the update operator
    under fp16 autocast per chunk, ONE ba over all edges with lm = 1e-5, ep = 1e-2), which is what configs[3] / [4] shard over 8 GPUs.
    Two steps (update_lowmem(steps=1) twice, a snapshot after each).  Two stand-ins of the oracle-backed droid_backends shim are
    switched on for this size (tests/golden/_shims_graph): the GEMM form of the alt-correlation (oracle.corr.altcorr_forward_fast,
    pinned to the per-tap form by a CPU test) and the per-pixel weight of the depth prior in `ba` (oracle.ba alpha_map; the
    reference's kernel has the constant 0.05, src/droid_kernels.cu:1405 -- with a constant map the two coincide).
    Kept per step: poses and per-frame means of depth / damping of every frame, depth and damping maps of C5_SAMPLE_FRAMES,
    target / weight / 4x4-subsampled hidden state of C5_SAMPLE_EDGES, per-edge means of |target - coords0|, weight, |hidden state|
    of ALL 8192 edges."""
    from golden_inputs import C5_SAMPLE_EDGES, C5_SAMPLE_FRAMES
    g = syn.make_graph("C5", with_features=True)
    if PROBE:
        g = perturb_one_ulp(g)
    N, ht, wd = g["n_frames"], g["ht"], g["wd"]
    video = ref_dv.DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, stereo=True, device="cpu")
    video.poses[:N] = torch.as_tensor(g["poses"]); video.disps[:N] = torch.as_tensor(g["disps"])
    video.intrinsics[:N] = torch.as_tensor(g["intrinsics"]); video.disps_sens[:N] = torch.as_tensor(g["disps_sens"])
    video.fmaps[:N] = torch.as_tensor(g["fmaps"]); video.nets[:N] = torch.as_tensor(g["nets"]); video.inps[:N] = torch.as_tensor(g["inps"])
    video.counter.value = N
    alpha = np.full((N + 2, ht, wd), 0.05, dtype=np.float32); alpha[:N] = g["disps_conf"]
    base.droid_backends.ALPHA_MAP, base.droid_backends.FAST_ALTCORR, base.droid_backends.BA_THREADS = alpha, True, 8
    out = {}
    try:
        with torch.no_grad():
            fg = ref_fg.FactorGraph(video, update_operator(1234), device="cpu", corr_impl="alt", max_factors=-1, upsample=False)
            fg.add_factors(torch.as_tensor(g["ii"]), torch.as_tensor(g["jj"]))
            assert len(fg.ii) == len(g["ii"]) == 8192
            out["ii"], out["jj"] = fg.ii.numpy().copy(), fg.jj.numpy().copy()
            sample = torch.as_tensor(C5_SAMPLE_EDGES); fr = np.asarray(C5_SAMPLE_FRAMES)
            for k in (1, 2):
                t = time.time()
                fg.update_lowmem(steps=1)
                print("C5 update_lowmem step %d: %.1f s" % (k, time.time() - t), flush=True)
                tag = "U%d" % k
                out[tag + "_poses"] = video.poses[:N].numpy().copy()
                d = video.disps[:N].numpy()
                out[tag + "_disps_f"] = d[fr].copy(); out[tag + "_disps_mean"] = d.reshape(N, -1).mean(1).copy()
                out[tag + "_damping_f"] = fg.damping[:N].numpy()[fr].copy()
                out[tag + "_damping_mean"] = fg.damping[:N].numpy().reshape(N, -1).mean(1).copy()
                tgt, wgt, net = fg.target[0], fg.weight[0], fg.net[0]
                out[tag + "_target_s"] = tgt[sample].numpy().copy(); out[tag + "_weight_s"] = wgt[sample].numpy().astype(np.float16)
                out[tag + "_net_s"] = net[sample][:, :, ::4, ::4].float().numpy().astype(np.float16)
                out[tag + "_flow_mean"] = (tgt - fg.coords0).abs().mean(dim=(1, 2, 3)).numpy().copy()
                out[tag + "_weight_mean"] = wgt.mean(dim=(1, 2, 3)).numpy().copy()
                out[tag + "_net_absmean"] = torch.stack([net[s:s + 256].float().abs().mean(dim=(1, 2, 3)) for s in range(0, net.shape[0], 256)]).reshape(-1).numpy().copy()
                np.savez_compressed("/tmp/graph_c5_%s_partial.npz" % ("probe" if PROBE else "golden"), **out)
    finally:
        base.droid_backends.ALPHA_MAP, base.droid_backends.FAST_ALTCORR, base.droid_backends.BA_THREADS = None, False, 1
    if PROBE:
        np.savez_compressed("/tmp/graph_c5_probe.npz", **out)
        return probe_report("C5", np.load(os.path.join(HERE, "graph_c5_python.npz")), out)
    np.savez_compressed(os.path.join(HERE, "graph_c5_python.npz"), **out)
    print("graph_c5: |dpose| %.3e %.3e" % (np.abs(out["U1_poses"] - g["poses"]).max(), np.abs(out["U2_poses"] - g["poses"]).max()))


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["stereo", "c2", "tum"]
    if "tum" in which:
        scenario_tum_size()
    if "wide" in which:
        scenario_wide()
    if "big" in which:
        scenario_big()
    if "stereo" in which:
        scenario_stereo()
    if "c2" in which:
        scenario_c2()
    if "c3" in which:                          # not in the default list: ~62 GB host RAM, tens of minutes
        scenario_c3()
    if "c5" in which:                          # not in the default list: ~25 GB host RAM, about an hour on 8 cores
        scenario_c5()
