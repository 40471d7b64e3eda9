#!/usr/bin/env python
"""Three FactorGraph.update iterations at C2 under `rocprofv3 --marker-trace --kernel-trace`: the ROCTX ranges of droid_amd/trace.py
(droid.update/{reproject,corr_lookup,update_operator,ba}) with the kernels launched inside each.

    rocprofv3 --marker-trace --kernel-trace --output-format csv -d OUT -o run -- python scripts/roctx_demo.py
    python scripts/roctx_demo.py --summarise OUT        # kernels per range from the two csv files
"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]


def summarise(out):
    mk = glob.glob(os.path.join(out, "**", "*marker_api_trace.csv"), recursive=True)
    kt = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)
    if not mk or not kt:
        print("no marker / kernel trace under", out); return 1
    ranges = [(r["Function"] if "Function" in r else r.get("Name", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(mk[0]))]
    ranges = [r for r in ranges if r[0].startswith("droid.")]
    kernels = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt[0]))]
    # a range is a HOST interval; kernels are attributed by the order of launches: those whose start falls between the range's
    # first and the next range's first kernel would need correlation ids -- here: host-side range durations + kernel count and
    # busy time of the kernels that START inside the host interval (launches are asynchronous, so this is a lower bound)
    agg = {}
    for name, a, b in ranges:
        ks = [k for k in kernels if a <= k[0] < b]
        n, host_us, busy = agg.get(name, (0, 0.0, 0.0))
        agg[name] = (n + 1, host_us + (b - a) / 1e3, busy + sum(k[1] - k[0] for k in ks) / 1e3)
    print("%-40s %6s %14s %22s" % ("range", "calls", "host us/call", "kernel us started inside"))
    for name, (n, h, busy) in sorted(agg.items()):
        print("%-40s %6d %14.1f %22.1f" % (name, n, h / n, busy / n))
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        sys.exit(summarise(sys.argv[2]))
    import numpy as np, torch
    from droid_amd import synthetic as syn, trace
    from droid_amd.depth_video import DepthVideo
    from droid_amd.factor_graph import FactorGraph
    from droid_amd.update import UpdateModule, empty_state_dict
    from droid_amd.weights import deterministic_state_dict

    class _SD:
        def state_dict(self):
            return empty_state_dict()
    assert trace.roctx_available(), "libroctx64 not found"
    g = syn.make_graph("C2", with_features=True)
    N, ht, wd = g["n_frames"], g["ht"], g["wd"]
    video = DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, device="cuda:0")
    d = lambda a: torch.as_tensor(a).cuda()
    video.poses[:N] = d(g["poses"]); video.disps[:N] = d(g["disps"]); video.intrinsics[:N] = d(np.tile(g["intrinsics"], (N, 1)))
    video.fmaps[:N] = d(g["fmaps"]); video.nets[:N] = d(g["nets"]); video.inps[:N] = d(g["inps"])
    video.counter.value = N
    upd = UpdateModule("cuda").load_state_dict(deterministic_state_dict(_SD(), seed=1234))
    fg = FactorGraph(video, upd, corr_impl="volume", max_factors=-1, upsample=False)
    fg.add_factors(d(g["ii"]), d(g["jj"]))
    for _ in range(4):
        fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
        torch.cuda.synchronize()
    with trace.HipTimer("update iteration at C2 (ms)"):
        fg.update(t0=1, t1=None, itrs=2, use_inactive=False)
    print("window spread of the flow:", fg.last_window_spread)
