#!/usr/bin/env python
"""TEST / MEASUREMENT INFRASTRUCTURE (not product): times the REFERENCE's own Python formulation of the hot path on the
host's cores -- the "Python/CPU fallback" BASELINE.json's north_star names:

  * geom/ba.py:31-106 `BA` + geom/chol.py:46-73 `schur_solve`  (dense Gauss-Newton step, itrs = 2)
  * modules/corr.py:23-50 `CorrBlock` (all-pairs volume + pooling, then the 4-level lookup)
  * droid_net.py:78-143 `UpdateModule` (+ modules/gru.py, GraphAgg)

imported UNMODIFIED from oracle/_ref/ref_py.zip (staged by oracle/build_ref.py; /root/reference itself is never read), on
BASELINE configs[1] = C2 (64 keyframes / 512 edges / 48x64), fp32, torch-CPU.  Its un-vendored dependencies are the stand-ins
of tests/golden/_shims (lietorch SE3 subset, torch_scatter); the reference has NO CPU implementation of
droid_backends.corr_index_forward (modules/corr.py:12 calls the CUDA extension), so that one call is served by the oracle's
torch-CPU restatement (oracle/corr.py:corr_block_lookup_torch's grid_sample form) -- stated in the output.

Runs in its own process (bench.py's cpu_baseline starts it) because the names it must bind -- `lietorch`, `torch_scatter`,
`droid_backends`, `geom`, `modules` -- are the product's module names inside bench.py.  Prints ONE JSON line.

    python oracle/time_reference_python.py [--threads T] [--lookup-edges 64] [--update-edges 128] [--repeats 3]
"""
import argparse
import json
import os
import statistics
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PYZIP = os.path.join(HERE, "_ref", "ref_py.zip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--lookup-edges", type=int, default=64)
    ap.add_argument("--update-edges", type=int, default=128)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--config", default="C2")
    a = ap.parse_args()
    if not os.path.exists(PYZIP):
        print(json.dumps({"skipped": "oracle/_ref/ref_py.zip not staged (oracle/build_ref.py needs /root/reference)"}))
        return
    sys.path[:0] = [os.path.join(ROOT, "tests", "golden", "_shims"), os.path.join(PYZIP, "droid_slam"), ROOT,
                    os.path.join(ROOT, "droid-slam_amd", "droid_amd")]
    import numpy as np
    import torch
    ncores = os.cpu_count() or 1
    threads = a.threads or min(ncores, 32)
    torch.set_num_threads(threads)

    import lietorch                                   # stand-in (tests/golden/_shims)
    import droid_backends                             # empty stand-in: modules/corr.py imports it at module scope
    assert "_shims" in lietorch.__file__ and "_shims" in droid_backends.__file__
    import geom.projective_ops as pops                # reference (from the archive)
    from geom.ba import BA                            # reference
    from modules.corr import CorrBlock                # reference
    import droid_net as ref_net                       # reference
    assert "ref_py.zip" in pops.__file__ and "ref_py.zip" in ref_net.__file__
    import synthetic as syn                           # droid_amd/synthetic.py (numpy only; the package __init__ needs the HIP extension)
    from oracle import corr as ocorr

    class _TorchProxy:                                # projective_ops.py:177 hard-codes device="cuda"
        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def as_tensor(*args, **kw):
            kw.pop("device", None)
            if not isinstance(args[0], torch.Tensor):
                kw.setdefault("dtype", torch.get_default_dtype())
            return torch.as_tensor(*args, **kw)
    pops.torch = _TorchProxy()

    def corr_index_forward(volume, coords, radius):   # what modules/corr.py:12 asks the CUDA extension for
        n, h, w = volume.shape[:3]
        out = ocorr.corr_block_lookup_torch([volume], coords.permute(0, 2, 3, 1), radius)       # [n,49,h,w]
        return [out.reshape(n, 2 * radius + 1, 2 * radius + 1, h, w)]
    droid_backends.corr_index_forward = corr_index_forward

    def timed(fn, reps):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    g = syn.make_graph(a.config, with_features=True)
    N, ht, wd, E = g["n_frames"], g["ht"], g["wd"], len(g["ii"])
    f32 = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32)
    ii, jj = torch.as_tensor(g["ii"]), torch.as_tensor(g["jj"])
    # ---- BA: two Gauss-Newton steps of geom/ba.py on the whole graph
    poses0 = lietorch.SE3(f32(g["poses"])[None]); disps0 = f32(g["disps"])[None]
    intr = f32(g["intrinsics"])[None, None].repeat(1, N, 1)
    target = f32(g["targets"]).permute(0, 2, 3, 1)[None].contiguous(); weight = f32(g["weights"]).permute(0, 2, 3, 1)[None].contiguous()
    kx = torch.unique(ii)
    kx_full = np.unique(np.concatenate([np.arange(1, N), g["ii"]]))
    eta = f32(g["eta"][np.searchsorted(kx_full, kx.numpy())])[None]

    def run_ba():
        p, d = poses0, disps0
        with torch.no_grad():
            for _ in range(g["itrs"]):
                p, d = BA(target, weight, eta, p, d, intr, ii, jj, fixedp=1)
        return p, d
    t_ba = timed(run_ba, a.repeats)
    # ---- correlation: build (once per edge lifetime, factor_graph.py:128-133) and lookup of a sample of edges
    nl = min(a.lookup_edges, E)
    fm = f32(g["fmaps"][:, 0].astype(np.float32))
    f1, f2 = fm[ii[:nl]][None], fm[jj[:nl]][None]
    yy, xx = torch.meshgrid(torch.arange(ht, dtype=torch.float32), torch.arange(wd, dtype=torch.float32), indexing="ij")
    rng = np.random.default_rng(0)
    coords = (torch.stack([xx, yy], -1)[None, None] + f32(rng.uniform(-6, 6, (1, nl, 1, 1, 2))) + 0.05 * yy[None, None, :, :, None]).contiguous()
    blk = [None]

    def build():
        with torch.no_grad():
            blk[0] = CorrBlock(f1, f2)
    t_build = timed(build, max(1, a.repeats - 1))

    def lookup():
        with torch.no_grad():
            return blk[0](coords)
    t_lk = timed(lookup, a.repeats)
    # ---- update operator (fp32) on a sample of edges
    nu = min(a.update_edges, E)
    m = ref_net.UpdateModule().eval()
    tg = torch.Generator().manual_seed(0)
    uargs = (torch.tanh(torch.randn(1, nu, 128, ht, wd, generator=tg)), torch.relu(torch.randn(1, nu, 128, ht, wd, generator=tg)),
             torch.randn(1, nu, 196, ht, wd, generator=tg), torch.randn(1, nu, 4, ht, wd, generator=tg), ii[:nu], jj[:nu])

    def update():
        with torch.no_grad():
            return m(*uargs)
    t_up = timed(update, a.repeats)
    HW = ht * wd
    per_ep = t_ba / (E * HW) + t_lk / (nl * HW) + t_up / (nu * HW)
    print(json.dumps({
        "value": 1.0 / per_ep, "unit": "edge-pixels/s", "kind": "reference", "config": a.config, "threads": threads, "host_cores": ncores,
        "dtype": "f32", "ba_s": t_ba, "lookup_s": t_lk, "update_s": t_up, "corr_build_s": t_build,
        "s_per_update_iteration": per_ep * E * HW,
        "sample": "the reference's own geom/ba.py:BA (itrs=%d, whole %s graph: %d keyframes / %d edges, %.2fs) + modules/corr.py:CorrBlock lookup of %d edges "
                  "(%.3fs; the volume build of those edges, %.2fs, is paid once per edge lifetime and not counted) + droid_net.py:UpdateModule on %d edges "
                  "(%.3fs), imported unmodified from oracle/_ref/ref_py.zip, torch-CPU fp32, %d threads, 1 warm-up + median of %d; lietorch / "
                  "torch_scatter = tests/golden/_shims; corr_index_forward (no CPU implementation in the reference) = the oracle's grid_sample form"
                  % (g["itrs"], a.config, N, E, t_ba, nl, t_lk, t_build, nu, t_up, threads, a.repeats)}))


if __name__ == "__main__":
    main()
