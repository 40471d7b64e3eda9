#!/usr/bin/env python
"""Benchmark of the dense-BA update operator on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one update iteration of the hot path over the whole synthetic frame graph
(BASELINE.json configs[2]: 512 keyframes / 4096 edges / 48x64, global-BA damping):
    reproject all edges -> 4-level correlation-pyramid lookup -> [ConvGRU update block] -> ba(itrs=2)
with every input resident in HBM before the timed region.  `value` = edges x pixels / second of the
whole job; `ms_per_global_ba` is the BA-only part (droid_backends.ba, itrs=2).

N > 1: one process per GPU (torch.distributed, RCCL); edges are sharded by source frame, the only
exchange is the all-reduce of the reduced camera system inside the BA ("multi-GPU").

The JSON line also carries `roofline` (correlation-lookup kernel, HBM bound, algorithmic bytes from
SURVEY.md 8d: 880 B per edge-pixel for fp16) and `cpu_baseline` (the numpy / torch-CPU oracle timed on all of this
host's cores: BA on the whole graph, lookup and update operator on bounded samples).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "droid-slam_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
if os.environ.get("DH_LIB_DIR"):          # same-box A/B of a compile-time variant of the library (droid-slam_amd/build.py DROID_HIP_VARIANT)
    sys.path.insert(0, os.path.join(PKG, os.environ["DH_LIB_DIR"]))

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0           # MI355X spec (MI355X_MICROARCH.md)
LOOKUP_BYTES_PER_EP_F16 = 880   # 2 * (240 taps + 196 samples) + 8 B of coordinates
LOOKUP_FUSED_BYTES_PER_EP_F16 = 744   # the lookup fused with the 196 -> 128 layer that reads it: 2 * (240 taps + 128 outputs) + 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-update-op", action="store_true", help="skip the ConvGRU block (diagnostics only)")
    ap.add_argument("--no-lookup", action="store_true", help="skip the correlation pyramid (BA-only diagnostics)")
    ap.add_argument("--nhwc-lookup", action="store_true", help="A/B: channel-last lookup output (7 pad channels per level) + "
                    "implicit-GEMM first correlation layer, instead of the reference-layout output + corr0_nchw")
    ap.add_argument("--unfused-lookup", action="store_true", help="A/B: the stand-alone lookup kernel (reference-layout output) + the "
                    "operator's own first correlation layer, instead of the fused kernel (corr_pyramid_lookup_corr0)")
    ap.add_argument("--per-edge-inp", action="store_true", help="context features gathered per edge and convolved with the "
                    "other 320 gate inputs (the reference's data flow) instead of once per source frame (A/B)")
    ap.add_argument("--op-chunks", type=int, default=0, help="run the update operator over this many contiguous groups of source "
                    "frames (0 = automatic: 1, or 2 when the pyramid leaves less than 70 GB of HBM -- C5 on one GPU)")
    ap.add_argument("--no-check", action="store_true", help="skip the untimed output check of the step (N = 1 only): the same "
                    "iteration through the reference-layout entry points (reference-layout volumes + corr_index_forward, the "
                    "reference interface of the update operator with per-edge context features, ba) from the same state")
    ap.add_argument("--no-lowmem", action="store_true", help="do not add the update_lowmem line (child process) to the default line")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic in this run (rocprofv3 --pmc pass in a child process)")
    ap.add_argument("--no-sensitivity", action="store_true", help="skip the untimed flow-sensitivity runs of the lookup kernel")
    ap.add_argument("--emulate-world", type=int, default=0, help="single-GPU emulation of ONE rank of an N-rank run: time the step on that "
                    "rank's edge shard (same poses, same redundant solve) WITHOUT the collectives -- the compute a rank would do; the "
                    "default line runs this for N = 2, 4, 8 in child processes and adds the modelled collectives (dist_projection)")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--no-projection", action="store_true", help="skip the untimed 1/2/4/8-rank projection (dist_projection: the BA timed on the "
                    "whole graph and on a 1/8 shard to separate its redundant part)")
    ap.add_argument("--lowmem", action="store_true", help="time FactorGraph.update_lowmem steps (the global-BA iteration, "
                    "reference factor_graph.py:266-330) instead of FactorGraph.update steps")
    ap.add_argument("--lowmem-corr", default="auto", choices=["auto", "alt", "pyramid"], help="correlation features of the "
                    "global-BA iteration: on-the-fly alt-correlation in source-frame chunks, or a pyramid built once per call")
    ap.add_argument("--chunk-frames", type=int, default=64, help="source frames per chunk of the alt-correlation path")
    ap.add_argument("--hand-rolled-dist", action="store_true", help="N > 1 through this script's own composition of the sharded step (rounds 2-5) "
                    "instead of the product class droid_amd.dist_graph.DistFactorGraph (A/B)")
    ap.add_argument("--no-glo-chain", action="store_true", help="A/B: the global-context reduction as its own pass over the hidden state at the start of "
                    "every step (rounds 1-5) instead of inside the previous step's q-gate launch")
    ap.add_argument("--no-product-class", action="store_true", help="skip the untimed `factor_graph_update` key (FactorGraph.update at the same size)")
    return ap.parse_args()


def measure_lookup_traffic(variant, edges, timeout_s=240):
    """HBM bytes per edge-pixel of the lookup kernel, measured now: `rocprofv3 --kernel-trace --pmc <TCC_EA0 request counters>`
    around scripts/bench_lookup.py (same kernel, same reprojection flow, `edges` edges), one pass -- the four counters fill
    the TCC's four slots (MI355X_MICROARCH.md "rocprofv3 PMC slots"); bytes = RDREQ_128B * 128 + (RDREQ - RDREQ_128B) * 64 +
    WRREQ_64B * 64 + (WRREQ - WRREQ_64B) * 32 -- request sizes, so the guide's x2 correction of FETCH_SIZE (which tallies a
    128-byte request at 64) is not needed.  -> (bytes per edge-pixel, source note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="dh_pmc_", dir="/tmp")
    flag = {"fused": ["--fused"], "nhwc": ["--nhwc"], "nchw": []}[variant]
    cmd = [rp, "--kernel-trace", "--pmc", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum",
           "--output-format", "csv", "-d", tmp, "-o", "run", "--", sys.executable, os.path.join(ROOT, "scripts", "bench_lookup.py"),
           "--edges", str(edges), "--reps", "2", "--flow", "reproj"] + flag
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s,
                           env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            lines = [l for l in r.stdout.splitlines() if "simple_timer" not in l and l.strip()]
            return None, "rocprofv3 --pmc pass failed (rc %d): %s" % (r.returncode, " | ".join(lines[-6:])[-600:])
        want = "pyr_lookup_corr0_kernel<64, 0" if variant == "fused" else "pyr_lookup_kernel"      # (<W, MODE = 0, MIX>: not the twin <64, 6, ..>)
        agg = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                if want in row["Kernel_Name"]:
                    n, v = agg.get(row["Counter_Name"], (0, 0.0))
                    agg[row["Counter_Name"]] = (n + 1, v + float(row["Counter_Value"]))
        c = {k: v / n for k, (n, v) in agg.items()}
        if not all(k in c for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")):
            return None, "counters missing in the rocprofv3 output: %s" % sorted(c)
        rd = c["TCC_EA0_RDREQ_128B_sum"] * 128 + (c["TCC_EA0_RDREQ_sum"] - c["TCC_EA0_RDREQ_128B_sum"]) * 64
        wr = c["TCC_EA0_WRREQ_64B_sum"] * 64 + (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"]) * 32
        return (rd + wr) / (edges * 48 * 64), ("measured in this run: rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum "
                                                 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -- scripts/bench_lookup.py --edges %d --flow reproj %s "
                                                 "(per-dispatch average of %d launches; read %.3f GB + written %.3f GB)" % (
                                                     edges, " ".join(flag), agg["TCC_EA0_RDREQ_sum"][0], rd / 1e9, wr / 1e9))
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 --pmc pass exceeded %d s" % timeout_s
    except Exception as exc:                                   # an informational field must never cost the bench line
        return None, "rocprofv3 --pmc pass: %r" % (exc,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def lowmem_line(timeout_s=300):
    """`bench.py --lowmem --steps 16 --warmup 1` in a child process -> its JSON line (ms_per_step of update_lowmem(steps=8) calls)"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--lowmem", "--steps", "16", "--warmup", "1"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"skipped": "bench.py --lowmem failed (rc %d): %s" % (r.returncode, r.stderr.strip()[-300:])}
        d = json.loads(lines[-1])
        return {k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "steps", "config", "ms_altcorr_per_step", "ms_pyramid_build_per_call")
                if k in d}
    except subprocess.TimeoutExpired:
        return {"skipped": "bench.py --lowmem exceeded %d s" % timeout_s}


def reference_python_baseline(threads, timeout_s=300):
    """The reference's OWN Python formulation (geom/ba.py:BA + modules/corr.py:CorrBlock + droid_net.py:UpdateModule,
    unmodified, from oracle/_ref/ref_py.zip) timed on this host at BASELINE configs[1] = C2, beside the port's C3 figure:
    north_star's "Python/CPU fallback".  Own process (oracle/time_reference_python.py): the reference binds the module
    names `lietorch`, `torch_scatter`, `droid_backends`, which are the product's modules in this one."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "time_reference_python.py"), "--threads", str(max(1, threads))]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s,
                           env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"skipped": "oracle/time_reference_python.py failed (rc %d): %s" % (r.returncode, r.stderr.strip()[-300:])}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"skipped": "oracle/time_reference_python.py exceeded %d s" % timeout_s}


def cpu_baseline(g, n_lookup_edges=64, n_update_edges=128, repeats=3):
    """Oracle (numpy / torch-CPU restatement of the reference kernels, kind "port") on this host's cores, on a bounded
    sample of the same workload; each leg of one update iteration (lookup, update operator, ba) is timed on its own sample
    and the legs are summed per edge-pixel.  Multi-core: the per-edge and per-depth-block stages of oracle.ba run on a
    thread pool (numpy releases the GIL), the lookup goes through torch's multi-threaded grid_sample, the update operator
    through torch's CPU convolutions, the solve through LAPACK; one warm-up run, then the median of `repeats`.
    BA sample = the WHOLE C3 graph (512 keyframes / 4096 edges, itrs = 2).  The single-core figure of the same legs is
    kept as `value_1core` (BA on the 128-frame sub-graph)."""
    from oracle import ba as oba, corr as ocorr, update as oupd
    from droid_amd.weights import deterministic_state_dict
    import statistics
    import threadpoolctl
    ncores = os.cpu_count() or 1
    ht, wd = g["ht"], g["wd"]
    rng = np.random.default_rng(0)

    def timed(fn, reps):
        fn()                                             # warm-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    # ---- samples
    f1 = torch.from_numpy(rng.standard_normal((n_lookup_edges, 128, ht * wd)).astype(np.float32)) / 4.0
    f2 = torch.from_numpy(rng.standard_normal((n_lookup_edges, 128, ht * wd)).astype(np.float32)) / 4.0
    vol = torch.matmul(f1.transpose(1, 2), f2).reshape(n_lookup_edges * ht * wd, 1, ht, wd)
    pyr = []
    for _ in range(4):
        pyr.append(vol.reshape(n_lookup_edges, ht, wd, vol.shape[-2], vol.shape[-1]))
        vol = torch.nn.functional.avg_pool2d(vol, 2, stride=2)
    coords = torch.from_numpy(np.stack([rng.uniform(0, wd, (n_lookup_edges, ht, wd)), rng.uniform(0, ht, (n_lookup_edges, ht, wd))], -1).astype(np.float32))

    class _SD:
        def state_dict(self):
            return oupd.empty_state_dict()
    sd = deterministic_state_dict(_SD(), seed=1234)
    tg = torch.Generator().manual_seed(0)
    E1 = n_update_edges
    uargs = (torch.randn(E1, 128, ht, wd, generator=tg), torch.randn(E1, 128, ht, wd, generator=tg),
             torch.randn(E1, 196, ht, wd, generator=tg), torch.randn(E1, 4, ht, wd, generator=tg), torch.arange(E1) // 8)

    def run_ba(sub, N, threads):
        poses = g["poses"][:N].astype(np.float32).copy(); disps = np.array(g["disps"][:N], dtype=np.float32, order="C")
        oba.ba(poses, disps, g["intrinsics"], g["disps_sens"][:N], sub["targets"], sub["weights"], sub["eta"], sub["ii"], sub["jj"],
               1, N, g["itrs"], g["lm"], g["ep"], False, dtype=np.float32, threads=threads, chunk=None if threads > 1 else 64)

    def subgraph(N):
        ii, jj = g["ii"], g["jj"]
        keep = (ii < N) & (jj < N)
        kx_full = np.unique(np.concatenate([np.arange(1, g["n_frames"]), ii]))
        kx = np.unique(np.concatenate([np.arange(1, N), ii[keep]]))
        return dict(ii=ii[keep], jj=jj[keep], targets=g["targets"][keep], weights=g["weights"][keep], eta=g["eta"][np.searchsorted(kx_full, kx)])

    def lookup():
        ocorr.corr_block_lookup_torch(pyr, coords, 3)

    def update():
        with torch.no_grad():
            oupd.update_forward(sd, *uargs)

    # ---- multi-core: every leg at the thread count where THIS host runs it fastest (probed on small samples: on the
    # 256-core bench host oneDNN convolutions and grid_sample peak at 8-16 threads and are 10-20x slower with 256)
    nthreads0 = torch.get_num_threads()
    cands = [c for c in (4, 8, 16, 32, 64, 128, 256) if c <= ncores] or [1]
    u8 = tuple(a[:8] for a in uargs); p8 = [v[:8] for v in pyr]; c8 = coords[:8]

    def probe(fn):
        best, best_t = cands[0], float("inf")
        for c in cands:
            torch.set_num_threads(c)
            fn()
            t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
            if dt < best_t:
                best, best_t = c, dt
            if dt > 3.0 * best_t:
                break                                    # far past the optimum: stop probing
        return best

    def upd8():
        with torch.no_grad():
            oupd.update_forward(sd, *u8)
    th_up = probe(upd8)
    th_lk = probe(lambda: ocorr.corr_block_lookup_torch(p8, c8, 3))
    th_ba = min(ncores, 64)
    full = subgraph(g["n_frames"])
    torch.set_num_threads(min(ncores, 32))               # LAPACK / BLAS inside the BA
    t_ba = timed(lambda: run_ba(full, g["n_frames"], th_ba), max(1, repeats - 1))
    torch.set_num_threads(th_lk)
    t_lk = timed(lookup, repeats)
    torch.set_num_threads(th_up)
    t_up = timed(update, repeats)
    ep_ba, ep_lk, ep_up = len(full["ii"]) * ht * wd, n_lookup_edges * ht * wd, E1 * ht * wd
    per_ep = t_ba / ep_ba + t_lk / ep_lk + t_up / ep_up
    # ---- one core (the figure of round 1): BA on the 128-frame sub-graph, 16-edge lookup / update samples
    torch.set_num_threads(1)
    with threadpoolctl.threadpool_limits(limits=1):
        n1 = min(128, g["n_frames"])                      # (C2 has 64 keyframes)
        sub = subgraph(n1)
        t0 = time.perf_counter(); run_ba(sub, n1, 1); t_ba1 = time.perf_counter() - t0
        small_pyr = [v[:16] for v in pyr]; small_c = coords[:16]
        t0 = time.perf_counter(); ocorr.corr_block_lookup_torch(small_pyr, small_c, 3); t_lk1 = time.perf_counter() - t0
        u16 = tuple(a[:16] for a in uargs)
        t0 = time.perf_counter()
        with torch.no_grad():
            oupd.update_forward(sd, *u16)
        t_up1 = time.perf_counter() - t0
    per_ep1 = t_ba1 / (len(sub["ii"]) * ht * wd) + t_lk1 / (16 * ht * wd) + t_up1 / (16 * ht * wd)
    torch.set_num_threads(nthreads0)
    ref_py = reference_python_baseline(th_up)
    ref_top = {}
    if isinstance(ref_py, dict) and "value" in ref_py:           # scalar keys at the top level (a parser that keeps only scalars keeps them)
        ref_top = {"reference_value": ref_py["value"], "reference_unit": ref_py.get("unit"), "reference_kind": "reference",
                   "reference_config": ref_py.get("config"), "reference_threads": ref_py.get("threads"),
                   "reference_s_per_update_iteration": ref_py.get("s_per_update_iteration"),
                   "reference_sample": ref_py.get("sample")}
    return {"reference_python_c2": ref_py, **ref_top,
            "value": 1.0 / per_ep, "unit": "edge-pixels/s", "cores": max(th_ba, th_lk, th_up), "kind": "port",
            "sample": "oracle (numpy/torch-CPU, fp32) on a %d-core host, every leg at its fastest thread count, 1 warm-up + median of %d: "
                      "ba itrs=%d on the whole %d-keyframe / %d-edge graph (%.2fs, %d threads) + 4-level lookup of %d edges (%.3fs, %d threads) "
                      "+ update operator on %d edges (%.3fs, %d threads)" % (
                          ncores, repeats, g["itrs"], g["n_frames"], len(full["ii"]), t_ba, th_ba, n_lookup_edges, t_lk, th_lk, E1, t_up, th_up),
            "host_cores": ncores, "threads": {"ba": th_ba, "lookup": th_lk, "update": th_up},
            "ba_s": t_ba, "lookup_s": t_lk, "update_s": t_up, "ms_per_global_ba_cpu": 1e3 * t_ba,
            "value_1core": 1.0 / per_ep1}


def project_ranks(db, g, dev, tensors, lk, up, ba):
    """PREDICTION, not a measurement (no multi-GPU node is reachable from the build sessions; the round driver's SCALE run is the
    measurement): the step time of the edge-sharded design at N = 1, 2, 4, 8 ranks, from quantities measured in THIS run:
      * lookup and update operator are per-edge work -> 1/N of this run's times (the C2 line, 512 edges = the per-rank share at
        N = 8, runs the operator at 6.35 ms against 49.0 / 8 = 6.1: +4 % small-batch penalty, applied for N > 1);
      * the BA splits into a part that shards with the edges (build, Gram, depth back-substitution) and a REDUNDANT part every
        rank repeats (damping, fp64 Cholesky, back solve, retraction): separated here by timing droid_backends.ba on the whole
        graph and on one rank's shard of 1/8 of the edges (same poses, same solve);
      * collectives per ba(): `itrs` all-reduces of the packed co-visible blocks + one of the depth maps, priced with the ring
        formula 2 (N-1)/N x bytes / link_rate + N-proportional hop latency.  link_rate = 100 GB/s (xGMI link 153.6 GB/s peak x 0.65:
        an ASSUMPTION until the driver's SCALE run), hop latency 6 us."""
    from droid_amd.dist_ba import shard_edges_by_source_frame, reduced_system_pattern
    poses0, disps0, intr, sens, tgt, wgt, eta, ii, jj, N = tensors
    E = int(ii.numel())
    shards, _ = shard_edges_by_source_frame(g["ii"], 8)
    mine = torch.as_tensor(shards[0], device=dev)

    def t_ba(sel):
        a = (tgt, wgt, ii, jj) if sel is None else (tgt[sel].contiguous(), wgt[sel].contiguous(), ii[sel].contiguous(), jj[sel].contiguous())

        def run():
            p, d = poses0.clone(), disps0.clone()
            db.ba(p, d, intr, sens, a[0], a[1], eta, a[2], a[3], 1, N, g["itrs"], g["lm"], g["ep"], False)
        return _time_ms(run, reps=5)
    t_full, t_shard = t_ba(None), t_ba(mine)
    frac = len(shards[0]) / float(E)
    shardable = max(0.0, (t_full - t_shard) / (1.0 - frac))
    redundant = max(0.0, t_full - shardable)
    bp, bq = reduced_system_pattern(g["ii"], g["jj"], 1, N)
    packed_bytes = 8 * (36 * len(bp) + 6 * (N - 1) + 2)
    disps_bytes = 4 * int(disps0.numel())
    link, hop = 100e9, 6e-6
    out = {"kind": "prediction from this run's single-GPU measurements (see bench.py project_ranks); NOT a measurement",
           "inputs": {"ms_corr_lookup": lk, "ms_update_operator": up, "ms_per_global_ba": ba, "ms_ba_whole_graph_isolated": t_full,
                      "ms_ba_one_eighth_shard_isolated": t_shard, "ms_ba_sharded_part": shardable, "ms_ba_redundant_part": redundant,
                      "allreduce_bytes_per_gn_iteration": packed_bytes, "allreduce_bytes_disps": disps_bytes,
                      "assumed_link_GBs": link / 1e9, "assumed_hop_latency_us": hop * 1e6, "small_batch_penalty": 1.04},
           "ranks": {}}
    for n in (1, 2, 4, 8):
        ar = lambda b: 0.0 if n == 1 else 1e3 * (2.0 * (n - 1) / n * b / link + 2 * (n - 1) * hop)
        coll = g["itrs"] * ar(packed_bytes) + ar(disps_bytes)
        pen = 1.0 if n == 1 else 1.04
        ms = pen * (lk + up) / n + shardable / n + redundant + coll
        out["ranks"][str(n)] = {"ms_per_step": ms, "ms_collectives": coll, "ms_ba": shardable / n + redundant + coll,
                                "speedup": (lk + up + shardable + redundant) / ms, "amdahl_redundant_share": redundant / ms}
    return out


def emulate_ranks(projection, config, timeout_s=150):
    """fills projection["ranks"][N]["ms_compute_measured"] (N = 2, 4, 8): `bench.py --emulate-world N --emulate-rank 0` in a child
    process = the step on rank 0's edge shard of an N-rank partition, same poses and the same redundant solve, no collectives;
    `ms_per_step_measured_compute_plus_modelled_collectives` adds the projection's collective model.  Any failure leaves the
    analytic figures alone."""
    import subprocess
    for n in (2, 4, 8):
        r = projection["ranks"].get(str(n))
        if r is None:
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--config", config, "--emulate-world", str(n), "--emulate-rank", "0", "--steps", "10",
               "--warmup", "3", "--no-cpu-baseline", "--no-check", "--no-lowmem", "--no-pmc", "--no-sensitivity", "--no-projection"]
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not lines:
                r["emulation"] = "failed (rc %d): %s" % (p.returncode, p.stderr.strip()[-200:])
                continue
            d = json.loads(lines[-1])
            r["ms_compute_measured"] = d["ms_per_step"]
            r["ms_compute_measured_parts"] = {k: d.get(k) for k in ("ms_corr_lookup", "ms_update_operator", "ms_per_global_ba")}
            r["edges_of_the_emulated_rank"] = d["emulated_rank"]["edges"]
            r["ms_per_step_measured_compute_plus_modelled_collectives"] = d["ms_per_step"] + r["ms_collectives"]
        except subprocess.TimeoutExpired:
            r["emulation"] = "exceeded %d s" % timeout_s
        except Exception as exc:
            r["emulation"] = repr(exc)
    one = projection["ranks"]["1"]["ms_per_step"]
    for n, r in projection["ranks"].items():
        if "ms_per_step_measured_compute_plus_modelled_collectives" in r:
            r["speedup_measured_compute"] = one / r["ms_per_step_measured_compute_plus_modelled_collectives"]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks here (one process per GPU, RCCL) by
    re-executing this command line under torch.distributed.run, the form the round driver itself uses for N > 1."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()
    backend = os.environ.get("DH_BENCH_BACKEND", "nccl")
    if have < n and backend != "gloo":
        sys.exit("bench.py --gpus %d: only %d device(s) visible; RCCL needs one GPU per rank (DH_BENCH_BACKEND=gloo runs "
                 "the multi-rank code path with several ranks per device, as a self-test)" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL between processes on this driver)
    sys.exit(subprocess.run(cmd, env=env).returncode)


_JSON_FD = None


def _emit(out):
    """the bench's one JSON line, on the process's real stdout"""
    line = json.dumps(out) + "\n"
    if _JSON_FD is None:
        sys.stdout.write(line); sys.stdout.flush()
    else:
        os.write(_JSON_FD, line.encode())


def _ev():
    return torch.cuda.Event(enable_timing=True)


def _lookup_rate(ms, E, HW, bytes_per_ep=LOOKUP_BYTES_PER_EP_F16):
    gbs = bytes_per_ep * E * HW / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"ms": ms, "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS}


def _time_ms(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = _ev(), _ev()
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def lookup_sensitivity(db, corr, g, coords_bench, ii, jj, fmaps, dev, lk_ms, upd=None):
    """The pyramid layout coalesces when the 64 pixels of an 8x8 source block want the same displacement cells, i.e. it is
    sensitive to the spatial coherence of the flow (the reference layout is not).  Same kernel, same pyramid, three flows:
      bench    the step's own coords (reprojection with the bench's initial state: depth 1 everywhere)
      planes   reprojection with the ground-truth poses and piecewise-constant depth planes (16x16-pixel patches, depth
               jumps between them): coherent inside a patch, discontinuous across
      random   independent uniform coordinates per pixel: no coherence at all (the worst case)
    and the reference-layout lookup (droid_backends.corr_index_forward on [n,h,w,h2,w2] volumes, 4 levels) on the first 256
    edges under the same three flows."""
    from droid_amd.corr import CorrBlockRef
    E, ht, wd = len(ii), g["ht"], g["wd"]
    HW = ht * wd
    rng = np.random.default_rng(11)
    d = lambda a, **kw: torch.as_tensor(np.ascontiguousarray(a), **kw).to(dev)
    N = g["n_frames"]
    planes = rng.uniform(0.3, 2.0, (N, ht // 16, wd // 16)).astype(np.float32)
    disps_planes = d(np.kron(planes, np.ones((16, 16), dtype=np.float32)))
    coords_planes, _ = db.reproject(d(g["poses_gt"]), disps_planes, d(g["intrinsics"]), ii, jj)
    coords_random = d(np.stack([rng.uniform(0, wd, (E, ht, wd)), rng.uniform(0, ht, (E, ht, wd))], -1).astype(np.float32))
    flows = {"bench": coords_bench, "planes": coords_planes, "random": coords_random}
    out = {}
    fused = upd is not None and bool(db.get_option("lookup_fused"))
    for name, c in flows.items():
        c = c.contiguous()
        ms = lk_ms if name == "bench" and lk_ms > 0 and not fused else _time_ms(lambda: corr(c[None]))
        out[name] = _lookup_rate(ms, E, HW)                 # the stand-alone lookup kernel (880 B/ep)
    if fused:                                               # the kernel of the timed step (744 B/ep)
        out["fused_kernel"] = {name: _lookup_rate(lk_ms if name == "bench" and lk_ms > 0 else _time_ms(lambda: corr.lookup_corr0(c.contiguous()[None], upd)),
                                                  E, HW, LOOKUP_FUSED_BYTES_PER_EP_F16) for name, c in flows.items()}
    n = min(256, E)
    rig = fmaps.shape[1]
    cidx = (ii[:n] == jj[:n]).long() if rig > 1 else torch.zeros_like(ii[:n])
    ref_block = CorrBlockRef(fmaps[ii[:n], 0][None], fmaps[jj[:n], cidx][None])
    refl = {"edges": n}
    for name, c in flows.items():
        cn = c[:n].contiguous()
        refl[name] = _lookup_rate(_time_ms(lambda: ref_block(cn[None])), n, HW)
    out["reference_layout_kernel"] = refl
    del ref_block
    return out


def check_step(db, upd, g, dev, state, product_step):
    """Untimed: the timed iteration again from a saved state, and the same iteration through the REFERENCE-LAYOUT entry points
    -- all-pairs volumes in the reference layout + droid_backends.corr_index_forward per level (CorrBlockRef = modules/corr.py:23-50),
    motion features in torch (factor_graph.py:221-222), the update operator through its reference interface with per-edge
    context features (448-channel gate convolutions), droid_backends.ba -- from the same poses / depths / hidden state."""
    from droid_amd.corr import CorrBlockRef
    poses0, disps0, net, ii, jj, fmaps, inps_frames, inp_index, target_prev, damping0, kx_t, uniq_ii, intr, sens, conf = state
    E, ht, wd, N = len(ii), g["ht"], g["wd"], g["n_frames"]
    net_saved = net.clone()
    pa, da = product_step()                                   # poses / disps after the product step (net updated in place)
    net_a = net.clone()
    net.copy_(net_saved)
    # ---- reference-layout iteration
    poses, disps = poses0.clone(), disps0.clone()
    coords1, _ = db.reproject(poses, disps, intr, ii, jj)
    rig = fmaps.shape[1]
    feats = torch.empty(E, 196, ht, wd, device=dev, dtype=torch.float16)
    for s in range(0, E, 128):
        e = slice(s, min(E, s + 128))
        cidx = (ii[e] == jj[e]).long() if rig > 1 else torch.zeros_like(ii[e])
        blk = CorrBlockRef(fmaps[ii[e], 0][None], fmaps[jj[e], cidx][None])
        feats[e] = blk(coords1[e][None])[0]
        del blk
    yy, xx = torch.meshgrid(torch.arange(ht, device=dev, dtype=torch.float32), torch.arange(wd, device=dev, dtype=torch.float32), indexing="ij")
    coords0 = torch.stack([xx, yy], -1)
    motn = torch.cat([coords1 - coords0, target_prev - coords1], -1).permute(0, 3, 1, 2).clamp(-64.0, 64.0)
    damping_buf = damping0.clone()
    delta = torch.empty(E, ht, wd, 2, device=dev); weight = torch.empty(E, ht, wd, 2, device=dev)
    net_b = torch.empty_like(net_saved)
    # (in groups of whole source frames: the reference interface materialises NCHW copies of everything)
    bounds = [0]
    iic = ii.cpu().numpy()
    for s in range(1024, E, 1024):
        while s < E and iic[s] == iic[s - 1]:
            s += 1
        if s < E and s > bounds[-1]:
            bounds.append(s)
    bounds.append(E)
    for a, b in zip(bounds[:-1], bounds[1:]):
        e = slice(a, b)
        n_out, dl, wt, eta_e, _ = upd.forward(net_saved[e].permute(0, 3, 1, 2)[None], inps_frames[inp_index[e]].permute(0, 3, 1, 2)[None],
                                              feats[e][None], motn[e][None], ii[e], jj[e])
        net_b[e] = n_out[0].permute(0, 2, 3, 1); delta[e] = dl[0]; weight[e] = wt[0]
        damping_buf[torch.unique(ii[e])] = eta_e[0]
    tgt = (coords1 + delta).permute(0, 3, 1, 2).contiguous(); wgt = weight.permute(0, 3, 1, 2).contiguous()
    eta_ba = (0.2 * damping_buf[kx_t] + 1e-7).contiguous()
    if conf is None:
        db.ba(poses, disps, intr, sens, tgt, wgt, eta_ba, ii, jj, 1, N, g["itrs"], g["lm"], g["ep"], False)
    else:
        db.ba_ex(poses, disps, intr, sens, conf, tgt, wgt, eta_ba, ii, jj, 1, N, g["itrs"], g["lm"], g["ep"], False)
    disps.clamp_(min=0.001)
    torch.cuda.synchronize()
    q, qr = pa[:, 3:].double(), poses[:, 3:].double()
    v = q[:, 3:4] * -qr[:, :3] + qr[:, 3:4] * q[:, :3] + torch.cross(q[:, :3], -qr[:, :3], dim=-1)
    rel = (da - disps).abs() / disps.abs().clamp(min=1.0)
    res = {"max_dtrans": float((pa[:, :3] - poses[:, :3]).abs().max()), "max_drot_rad": float(2 * v.norm(dim=-1).max()),
           "disps_rel_q99": float(torch.quantile(rel.flatten()[:: max(1, rel.numel() // 1000000)], 0.99)), "disps_rel_max": float(rel.max()),
           "hidden_state_max_abs_diff": float((net_a.float() - net_b.float()).abs().max()),
           "pose_update_norm": float((pa[:, :3] - poses0[:, :3]).abs().max()),
           "against": "SELF-CONSISTENCY, not the oracle (that is tests/test_scale_gpu.py::test_composed_update_at_c3_matches_reference_factor_graph): this "
                      "library's reference-layout entry points (reference-layout volumes + corr_index_forward per level, UpdateModule.forward with "
                      "per-edge context features, ba) from the same state"}
    # thresholds = 10 x what this line measures at C3 / C5 (poses 1.2e-7 .. 2e-7, depths q99 1.5e-5, hidden state 2^-11 = one fp16 ulp
    # below 1): a regression of the product path against its own reference-layout entry points shows here
    res["thresholds"] = {"max_dtrans": 2e-6, "max_drot_rad": 2e-6, "disps_rel_q99": 2e-4, "hidden_state_max_abs_diff": 2.0 ** -8}
    res["ok"] = bool(all(res[k] <= v for k, v in res["thresholds"].items()) and np.isfinite(res["max_dtrans"]) and res["pose_update_norm"] > 0)
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (no CPU fallback)"
    local = local % torch.cuda.device_count()            # (several ranks on one device only in the gloo self-test below)
    torch.cuda.set_device(local)
    # DH_BENCH_DIST1=1: the multi-rank code path (process group, DistBA, collectives, `dist` record) with ONE rank -- on the `nccl`
    # backend this is RCCL executing the edge-sharded step on a single-GPU box
    dist_on = world > 1 or os.environ.get("DH_BENCH_DIST1", "0") == "1"
    if dist_on:
        # RCCL and gloo write banners ("RCCL version : ...", "[Gloo] Rank 0 is connected ...") to the C-level stdout, in no defined order
        # with Python's: the process's stdout carries the ONE JSON line and nothing else -- everything else goes to stderr
        global _JSON_FD
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL ("nccl") is the product path; DH_BENCH_BACKEND=gloo exists to exercise the multi-rank code path on a box
        # with a single GPU (RCCL refuses two ranks on one device)
        dist.init_process_group(os.environ.get("DH_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    if args.lowmem:
        return main_lowmem(args, dev, world, rank)
    if dist_on and not args.hand_rolled_dist:
        return main_dist(args, dev, world, rank, local)

    import droid_backends as db
    from droid_amd import synthetic as syn
    from droid_amd.corr import CorrBlock

    cfg = syn.CONFIGS[args.config]
    g = syn.make_graph(cfg, with_features=True)
    N, ht, wd = g["n_frames"], g["ht"], g["wd"]
    HW = ht * wd
    # edges grouped by source frame (a permutation of the seeded edge list; FactorGraph appends edges in that order too,
    # factor_graph.py:332-344): every rank's shard and every operator chunk is then a contiguous range of records
    order = np.argsort(g["ii"], kind="stable")
    for k in ("ii", "jj", "targets", "weights"):
        g[k] = g[k][order]
    ii_all, jj_all = g["ii"], g["jj"]
    E_all = len(ii_all)

    # ---- edge sharding by source frame (contiguous frame ranges balanced by edge count) ----
    from droid_amd.dist_ba import DistBA, shard_edges_by_source_frame, local_eta_rows
    emulate = args.emulate_world > 1 and world == 1
    shard_world, shard_rank = (args.emulate_world, args.emulate_rank) if emulate else (world, rank)
    assert 0 <= shard_rank < shard_world
    shards, bounds = shard_edges_by_source_frame(ii_all, shard_world)
    mine = shards[shard_rank]
    eta_rows, kx_local = local_eta_rows(ii_all, ii_all[mine], 1, N)      # depth blocks of this rank's BA
    d = lambda a, **kw: torch.as_tensor(np.ascontiguousarray(a), **kw).to(dev)
    ii, jj = d(ii_all[mine]), d(jj_all[mine])
    E = len(mine)

    poses0, disps0 = d(g["poses"]), d(g["disps"])
    poses, disps = poses0.clone(), disps0.clone()
    intr, sens, eta = d(g["intrinsics"]), d(g["disps_sens"]), d(g["eta"][eta_rows])
    # BASELINE configs[4]: "per-pixel depth-confidence weights" -- a seeded NON-constant weight of the sensor-depth prior
    # (droid_amd.synthetic.depth_confidence) through droid_backends.ba_ex / DistBA.ba(alpha=...); None for the configs without sensor depth
    conf = d(g["disps_conf"]) if "disps_conf" in g else None
    kx_t = d(kx_local)
    damping_buf = torch.full((N, ht, wd), 1e-6, device=dev, dtype=torch.float32)     # factor_graph.py:56
    targets, weights = d(g["targets"][mine]), d(g["weights"][mine])
    rig = g["fmaps"].shape[1]
    fmaps = d(g["fmaps"])

    # ---- correlation pyramid for this rank's edges (setup, untimed: built once per edge lifetime) ----
    c = (ii == jj).long() if rig > 1 else torch.zeros_like(ii)
    ms_build = ms_alloc = ms_rebuild = None
    if args.no_lookup:
        corr = None
    else:
        CorrBlock.from_frames(fmaps, ii[:8], jj[:8])                             # code objects, LDS opt-in, allocator
        torch.cuda.synchronize()
        # the pyramid's STORAGE first, timed on its own with the host clock: hipMalloc of 105 GB at C3 maps the pages and is
        # synchronous -- seconds on a cold device (the first process of a fresh box), not part of the build
        t_a = time.perf_counter()
        arena = CorrBlock.arena(E, ht, wd, dev)
        torch.cuda.synchronize()
        ms_alloc = 1e3 * (time.perf_counter() - t_a)
        e0, e1, e2 = _ev(), _ev(), _ev()
        # the product's build (FactorGraph.add_factors / update_lowmem): features transposed + pooled once per frame, the build
        # kernel indexed by the edges' frames -- all of this rank's edges
        e0.record(); corr = CorrBlock.from_frames(fmaps, ii, jj, out=arena); e1.record()        # first build into fresh pages
        corr = CorrBlock.from_frames(fmaps, ii, jj, out=arena); e2.record()                     # the same build again (what update_lowmem pays per call)
        torch.cuda.synchronize()
        ms_build = e0.elapsed_time(e1)
        ms_rebuild = e1.elapsed_time(e2)
        del arena
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                              # (the per-edge feature gathers and the build workspace)
    # ---- ConvGRU update operator: random-init weights of the reference architecture (no droid.pth here) ----
    upd = None
    op_chunks = 1
    if not args.no_update_op:
        from droid_amd.update import UpdateModule, empty_state_dict
        from droid_amd.weights import deterministic_state_dict

        class _SD:
            def state_dict(self):
                return empty_state_dict()
        upd = UpdateModule(dev).load_state_dict(deterministic_state_dict(_SD(), seed=1234))
        net = d(g["nets"])[ii].permute(0, 2, 3, 1).contiguous()        # hidden state per edge, NHWC fp16
        # context features: frame-level table of this rank's source frames (the reference gathers video.inps[ii] per edge,
        # factor_graph.py:135; every edge of a source frame carries the same 128 channels)
        f_lo, f_hi = int(bounds[shard_rank]), int(min(bounds[shard_rank + 1], N))
        inps_frames = d(g["inps"])[f_lo:f_hi].permute(0, 2, 3, 1).contiguous()       # [frames,h,w,128] f16
        inp_index = (ii - f_lo).contiguous()
        inp_edges = inps_frames[inp_index].contiguous() if args.per_edge_inp else None
        # the operator's activations take ~3.5 KB per edge-pixel: with the 210 GB pyramid of C5 on one GPU it runs over two
        # groups of source frames (GraphAgg averages over the edges of a source frame: groups keep them together)
        free_b = torch.cuda.mem_get_info(dev)[0]
        op_chunks = args.op_chunks if args.op_chunks > 0 else (1 if free_b > 1.25 * E * HW * 3584 else max(2, int(np.ceil(1.6 * E * HW * 3584 / free_b))))
    iic = ii_all[mine]
    cuts = [0]
    for k in range(1, op_chunks):
        s = (E * k) // op_chunks
        while s < E and iic[s] == iic[s - 1]:
            s += 1
        if cuts[-1] < s < E:
            cuts.append(s)
    cuts.append(E)
    chunk_slices = [slice(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    ii_chunks = [ii[s].contiguous() for s in chunk_slices]                 # (kept: the operator caches per edge-list tensor)
    idx_chunks = [inp_index[s].contiguous() for s in chunk_slices] if upd is not None else None
    uniq_chunks = [torch.unique(t) for t in ii_chunks]
    target_prev = targets.permute(0, 2, 3, 1).contiguous()                # [E,h,w,2]
    torch.cuda.synchronize()

    solver = DistBA(world, always_reduce=(world == 1)) if dist_on else None
    if solver is not None:
        solver.set_owned_frames(bounds[rank], bounds[rank + 1])
        solver.set_graph(ii_all, jj_all, 1, N)              # all-reduce of the co-visible 6x6 blocks only (~6 MB instead of 77 MB)
    uniq_ii = torch.unique(ii)

    lk_ms, up_ms, ba_ms = [], [], []
    # the ConvGRU's global-context sums of the hidden state a step writes, computed inside its q-gate launch and consumed by the next step
    # on the same tensor -- what FactorGraph._operator does between update iterations (droid_amd/factor_graph.py); --no-glo-chain: every
    # step starts with the stand-alone reduction over the hidden state, as in rounds 1-5
    glo_chain = not args.no_glo_chain
    glo_box = [None]
    ctx_box = [None]       # set only for the extra `steady_state_cached_context` loop after the timed region (see below)

    # the update operator takes the unpadded reference-layout features where its first layer has the kernel for them
    ref_layout = upd is not None and upd.wants_reference_layout_corr(ht, wd) and not args.nhwc_lookup
    # default: the lookup runs fused with the first layer of the operator's correlation encoder (option lookup_fused)
    fused_lookup = upd is not None and corr is not None and bool(db.get_option("lookup_fused")) and not args.nhwc_lookup and not args.unfused_lookup

    def step(timed):
        """one FactorGraph.update iteration (reference factor_graph.py:214-263)"""
        poses.copy_(poses0); disps.copy_(disps0)
        coords1, _ = db.reproject(poses, disps, intr, ii, jj)              # [E,h,w,2]
        e0, e1, e2, e3 = _ev(), _ev(), _ev(), _ev()
        e0.record()
        fused = upd is not None and corr is not None
        # channel-last features straight into the update operator; the reference-layout [E,196,h,w] otherwise
        corr0 = feats = None
        if fused_lookup:
            corr0 = corr.lookup_corr0(coords1[None], upd)                     # [E,h,w,128]: lookup + Conv2d(196,128,1) + ReLU
        elif corr is not None:
            feats = corr.lookup_nhwc(coords1[None]) if (fused and not ref_layout) else corr(coords1[None])[0]
        e1.record()
        if upd is not None and (feats is not None or corr0 is not None):
            flow = db.motion_features(coords1, target_prev)                   # factor_graph.py:221-222
            if len(chunk_slices) == 1:
                if inp_edges is not None:
                    _, _, _, damping, upmask = upd.forward_nhwc(net, inp_edges, feats, flow, ii, corr0=corr0, glo_red=glo_box[0], glo_next=glo_chain)
                else:
                    _, _, _, damping, upmask = upd.forward_nhwc(net, None, feats, flow, ii, inp_frames=inps_frames, inp_index=inp_index, corr0=corr0,
                                                                ctx=ctx_box[0], glo_red=glo_box[0], glo_next=glo_chain)
                glo_box[0] = upd.last_glo                                    # (None where the q gate's launch cannot reduce the new state)
                dw = upd.last_dw
                damping_buf[uniq_ii] = damping                               # factor_graph.py:238
            else:
                dw = torch.empty(E, ht, wd, 4, device=dev, dtype=torch.float32)
                for s, iis, ixs, uq in zip(chunk_slices, ii_chunks, idx_chunks, uniq_chunks):
                    f_s = None if feats is None else (feats[:, s] if feats.dim() == 5 else feats[s])
                    c_s = None if corr0 is None else corr0[s]
                    if inp_edges is not None:
                        _, _, _, damping, upmask = upd.forward_nhwc(net[s], inp_edges[s], f_s, flow[s], iis, corr0=c_s)
                    else:
                        _, _, _, damping, upmask = upd.forward_nhwc(net[s], None, f_s, flow[s], iis, inp_frames=inps_frames, inp_index=ixs, corr0=c_s,
                                                                    ctx=ctx_box[0])
                    dw[s] = upd.last_dw
                    damping_buf[uq] = damping
            _, _, tgt, wgt = db.ba_inputs(coords1, dw)                        # target = coords1 + delta, [E,2,h,w] for ba (:233,253-254)
            eta_ba = (0.2 * damping_buf[kx_t] + 1e-7).contiguous()            # factor_graph.py:251 (rows = depth blocks)
        else:
            tgt, wgt, eta_ba = targets, weights, eta
        e2.record()
        if solver is not None:
            solver.ba(poses, disps, intr, sens, tgt, wgt, eta_ba, ii, jj, 1, N, g["itrs"], g["lm"], g["ep"], alpha=conf)
        elif conf is not None:
            db.ba_ex(poses, disps, intr, sens, conf, tgt, wgt, eta_ba, ii, jj, 1, N, g["itrs"], g["lm"], g["ep"], False)
        else:
            db.ba(poses, disps, intr, sens, tgt, wgt, eta_ba, ii, jj, 1, N, g["itrs"], g["lm"], g["ep"], False)
        disps.clamp_(min=0.001)
        e3.record()
        if timed:
            lk_ms.append((e0, e1)); up_ms.append((e1, e2)); ba_ms.append((e2, e3))
        return coords1

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not part of the W warm-up steps: the first two passes load the code objects, opt kernels into >64 KB of LDS
    # and let torch's caching allocator reach its steady-state pool (its hipMallocs synchronise the device)
    for _ in range(2):
        step(False)
    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    lk = float(np.mean([a.elapsed_time(b) for a, b in lk_ms])) if lk_ms else 0.0
    up = float(np.mean([a.elapsed_time(b) for a, b in up_ms])) if up_ms else 0.0
    ba = float(np.mean([a.elapsed_time(b) for a, b in ba_ms])) if ba_ms else 0.0

    # Outside the timed region and NOT part of `value`: the same step with the gates' per-frame context term (UpdateModule.context_term:
    # the convolution over the 128 context channels, which depend on the keyframes only) computed once instead of in every
    # step -- what droid_amd.factor_graph.FactorGraph.update does between keyframe insertions (its `_context` cache).  The timed
    # steps above recompute it every time, as the reference's convolutions over [net, inp, corr, flow] do.
    steady = None
    if world == 1 and not emulate and upd is not None and corr is not None and inp_edges is None and ht % 4 == 0 and wd == 64:
        ctx_box[0] = upd.context_term(inps_frames)
        for _ in range(2):
            step(False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(False)
        torch.cuda.synchronize()
        steady = {"ms_per_step": 1000.0 * (time.perf_counter() - t1) / max(1, args.steps),
                  "note": "context term of the gate convolutions computed once (FactorGraph.update's cache); not the headline value"}
        ctx_box[0] = None

    projection = None
    if world == 1 and not emulate and upd is not None and corr is not None and not args.no_projection and cfg.name in ("C3", "C5"):
        try:
            # (the graph's own synthetic BA inputs: ground-truth reprojection + noise as targets, weights in (0, 1))
            projection = project_ranks(db, g, dev, (poses0, disps0, intr, sens, targets, weights, eta, ii, jj, N), lk, up, ba)
        except Exception as exc:                          # an informational field must never cost the bench line
            projection = {"error": repr(exc)}
    check = sens_out = None
    if world == 1 and not emulate and upd is not None and corr is not None and inp_edges is None:
        if not args.no_check:
            def product():
                step(False)
                torch.cuda.synchronize()
                return poses.clone(), disps.clone()
            try:
                check = check_step(db, upd, g, dev, (poses0, disps0, net, ii, jj, fmaps, inps_frames, inp_index, target_prev,
                                                     torch.full((N, ht, wd), 1e-6, device=dev), kx_t, uniq_ii, intr, sens, conf), product)
            except torch.cuda.OutOfMemoryError as exc:        # C5 on one GPU: no room for the reference-layout copies
                check = {"ok": None, "skipped": "out of memory for the reference-layout run: %s" % str(exc)[:80]}
            glo_box[0] = None                                 # (check_step restored the hidden state in place: the kept sums belong to another state)
            torch.cuda.empty_cache()
        if not args.no_sensitivity:
            coords_bench, _ = db.reproject(poses0, disps0, intr, ii, jj)
            try:
                sens_out = lookup_sensitivity(db, corr, g, coords_bench, ii, jj, fmaps, dev, lk, upd if fused_lookup else None)
            except torch.cuda.OutOfMemoryError as exc:
                sens_out = {"skipped": "out of memory: %s" % str(exc)[:80]}

    dist_info = None
    if dist_on:                                                 # auditable record of the process group behind a multi-GPU line
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "device": torch.cuda.get_device_name(local), "local_rank": local,
                                          "edges": int(E), "frames": [int(bounds[rank]), int(min(bounds[rank + 1], N))]})
        # what the collectives of one ba() cost on this transport, measured on their own (median of 5, every rank takes part):
        # `itrs` all-reduces of the packed system + one of the depth maps; ms_per_global_ba minus this is the ranks' own work
        def _ar_ms(t):
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter(); dist.all_reduce(t); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
            return sorted(ts)[2]
        ar_sys = _ar_ms(torch.zeros(max(1, int(solver.last_exchange_bytes) // 8), dtype=torch.float64, device=dev))
        ar_disps = _ar_ms(torch.zeros_like(disps0))
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": per_rank,
                     "allreduce_bytes_per_gn_iteration": int(solver.last_exchange_bytes), "packed_exchange": bool(solver.last_exchange_packed),
                     "ms_allreduce_system": ar_sys, "ms_allreduce_disps": ar_disps,
                     "ms_collectives_per_global_ba": g["itrs"] * ar_sys + ar_disps}
    if rank == 0:
        ms = 1000.0 * elapsed / max(1, args.steps)
        ep_total = E_all * HW
        # (the channel-last variant also writes 7 zero channels per level: 936 B/ep physical; only the 880 B count)
        lookup_bytes = (LOOKUP_FUSED_BYTES_PER_EP_F16 if fused_lookup else LOOKUP_BYTES_PER_EP_F16) * E * HW
        achieved = lookup_bytes / (lk * 1e-3) / 1e9 if lk > 0 else 0.0
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc, separate passes, gfx950
        # FETCH_SIZE correction; scripts/pmc_bench_lookup.sh), scaled to this rank's edge-pixels; None if not measured
        traffic = traffic_src = None
        variant = "fused" if fused_lookup else "nhwc" if (upd is not None and corr is not None and not ref_layout) else "nchw"
        try:
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_lookup_pmc.json"))
            pmc = json.load(open(os.path.join(ROOT, "profiles", cands[-1])))             # the latest round's passes
            traffic = pmc[variant]["hbm_bytes_per_edge_pixel"] * E * HW
            traffic_src = "profiles/%s (rocprofv3 --pmc passes of the same kernel at 4096 edges, committed; NOT re-measured in this run)" % cands[-1]
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "BA update iterations/sec (edges*pixels/s), 512-KF graph",
            "value": ep_total / (ms * 1e-3), "unit": "edge-pixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (BA; fp64 solve) / f16 (correlation pyramid)", "data": "synthetic",
            "config": {"workload": "%s: %d keyframes, %d edges, %dx%d, ba itrs=%d lm=%g ep=%g%s" % (
                cfg.name, N, E_all, ht, wd, g["itrs"], g["lm"], g["ep"],
                ", stereo + sensor depth with per-pixel confidence weights (seeded, non-constant)" if cfg.stereo else ""),
                "stages": "reproject + corr lookup (4 levels, materialised fp16 pyramid, MI355X layout%s) + %sba" % (
                    ", fused with the correlation encoder's first layer" if fused_lookup else "",
                    "ConvGRU update operator (random-init weights) + " if upd is not None else ""),
                "parallelism": "edge-sharded x%d" % world, "update_operator_chunks": len(chunk_slices),
                "pyramid_GB": (corr.bytes() / 1e9 if corr is not None else 0.0)},
            "ms_per_global_ba": ba, "ms_corr_lookup": lk, "ms_update_operator": up,
            # setup, outside the timed steps: once per edge lifetime (factor_graph.py:128-133).  alloc = the storage (hipMalloc,
            # host clock); the two builds: HIP events
            # ms_pyramid_build = the build into storage that has been written before (what every build after a process's first one
            # costs: box-independent); ms_pyramid_first_build = the first one, which also touches 105 GB of fresh pages (61-77 ms on a
            # fresh box); ms_pyramid_rebuild = alias of ms_pyramid_build (the key of the profiles/r05_* files, where ms_pyramid_build
            # still was the first build)
            "ms_pyramid_alloc": ms_alloc, "ms_pyramid_first_build": ms_build, "ms_pyramid_build": ms_rebuild, "ms_pyramid_rebuild": ms_rebuild,
            "roofline": {"kernel": "pyr_lookup_corr0_kernel<64> (1 launch = 4-level pyramid lookup of all edges + the 196 -> 128 layer that "
                         "consumes it; 744 B/ep = 2*(240 taps + 128 outputs) + 8)" if fused_lookup else
                         "pyr_lookup_kernel<64, %s> (1 launch = 4-level pyramid lookup of all edges)" % (
                "channel-last" if (upd is not None and corr is not None and not ref_layout) else "reference layout"), "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch_group": lookup_bytes},
        }
        if fused_lookup and lk > 0:
            # what the same launch replaces, for comparison with earlier rounds' lines: the stand-alone lookup's 880 B/ep
            # (its 392 B/ep of samples now stay in registers) over the fused kernel's duration -- work-equivalent, NOT bytes moved
            out["roofline"]["standalone_lookup_equivalent_GBs"] = LOOKUP_BYTES_PER_EP_F16 * E * HW / (lk * 1e-3) / 1e9
        if steady is not None:
            out["steady_state_cached_context"] = steady
        if sens_out is not None:
            out["roofline_sensitivity"] = sens_out
        if check is not None:
            out["check"] = check
        if dist_info is not None:
            out["dist"] = dist_info
        if projection is not None:
            out["dist_projection"] = projection
        if upd is not None and up > 0:
            # secondary roofline (the contract's `roofline` object stays the HBM-bound lookup named by north_star): the
            # update operator is where the step time goes; algorithmic flops of this rank's convolutions (edge-level ones
            # per edge-pixel, GraphAgg's conv2 / eta / upmask per frame-pixel) over the live HIP-event time of the operator
          try:
            from droid_amd.update import PARAM_SHAPES
            frame_level = ("agg.conv2", "agg.eta.0", "agg.upmask.0")
            glo_vectors = ("gru.convz_glo", "gru.convr_glo", "gru.convq_glo")          # 1x1 on one vector per edge
            macs = lambda names: sum(co * ci * k * k for n, (co, ci, k) in PARAM_SHAPES.items() if n in names)
            edge_macs = macs([n for n in PARAM_SHAPES if n not in frame_level + glo_vectors])
            flops = 2.0 * (edge_macs * E * HW + macs(frame_level) * len(uniq_ii) * HW + macs(glo_vectors) * E)
            tf = flops / (up * 1e-3) / 1e12
            out["roofline_update_operator"] = {"kernel": "conv3x3_halo2_kernel / conv3x3_halo_kernel / conv_igemm_kernel (all of one "
                                               "UpdateModule.forward)", "bound": "mfma", "achieved": tf, "peak": 2500.0,
                                               "unit": "TFLOP/s", "frac": tf / 2500.0, "algorithmic_flops": flops}
          except Exception as exc:                        # an informational field must never cost the bench line
            out["roofline_update_operator"] = {"error": repr(exc)}
        if emulate:
            out["emulated_rank"] = {"rank": shard_rank, "of": shard_world, "edges": int(E), "frames": [f_lo, f_hi] if upd is not None else None,
                                    "note": "ONE process on ONE GPU running the step on this rank's edge shard: the rank's compute (lookup, "
                                            "update operator, build of its edges, the redundant solve) WITHOUT the collectives; `value` "
                                            "counts this shard's edge-pixels only; not a multi-GPU measurement"}
            out["value"] = E * HW / (ms * 1e-3)
            _emit(out)
            return
        if world == 1 and not args.no_product_class and upd is not None and corr is not None and inp_edges is None and len(chunk_slices) == 1:
            # the PRODUCT CLASS at the same size: droid_amd.factor_graph.FactorGraph.update (what factor_graph.py / the frontend call),
            # wall clock over `steps` iterations; its own pyramid, so this process's is released first
            corr.pyramid = None
            corr = None
            import gc
            gc.collect(); torch.cuda.empty_cache()
            try:
                out["factor_graph_update"] = product_class_line(args, g, dev, upd, conf, None)
            except Exception as exc:                        # an informational field must never cost the bench line
                out["factor_graph_update"] = {"error": repr(exc)[:300]}
            gc.collect(); torch.cuda.empty_cache()
        if not args.no_cpu_baseline and world == 1:        # rank 0 at N = 1 only (the other ranks would wait in teardown)
            out["cpu_baseline"] = cpu_baseline(g)
        had_corr = not args.no_lookup
        if world == 1 and not args.no_lowmem and had_corr and cfg.name == "C3":
            # the global-BA iteration (FactorGraph.update_lowmem, factor_graph.py:266-330) on the same graph, as an extra key of
            # the default line: own process (`bench.py --lowmem`), after this one's pyramid is released
            if corr is not None:
                corr.pyramid = None
            import gc
            gc.collect(); torch.cuda.empty_cache()
            out["lowmem"] = lowmem_line()
        if world == 1 and not args.no_pmc and had_corr:
            # roofline.traffic measured IN THIS RUN: one rocprofv3 --pmc pass over the same kernel / flow / edge count in a
            # child process, after everything is timed and the bench's own pyramid (105 GB at C3) is released
            if corr is not None:
                corr.pyramid = None
            import gc
            gc.collect(); torch.cuda.empty_cache()
            # (bytes per edge-pixel do not depend on the edge count once the pyramid is far beyond the caches; 1024 edges =
            # 26 GB keep the child process clear of whatever this process still holds)
            per_ep, why = measure_lookup_traffic(variant, min(E, 1024))
            if per_ep is not None:
                out["roofline"]["traffic"], out["roofline"]["traffic_source"] = per_ep * E * HW, why
            elif out["roofline"]["traffic_source"]:
                out["roofline"]["traffic_source"] += " [in-run pass: %s]" % why
        if projection is not None and "ranks" in projection and had_corr:
            # the per-rank COMPUTE of the projection measured instead of scaled: child processes run one rank's shard of an
            # N-rank partition on this GPU (bench.py --emulate-world N); the collectives stay modelled
            if corr is not None:
                corr.pyramid = None
            import gc
            gc.collect(); torch.cuda.empty_cache()
            emulate_ranks(projection, cfg.name)
        _emit(out)
    if dist_on:
        dist.destroy_process_group()


def _video_for(g, cfg, dev):
    """droid_amd.depth_video.DepthVideo holding the synthetic graph's frames (replicated on every rank)"""
    from droid_amd.depth_video import DepthVideo
    N, ht, wd = g["n_frames"], g["ht"], g["wd"]
    d = lambda a, **kw: torch.as_tensor(np.ascontiguousarray(a), **kw).to(dev)
    video = DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, stereo=cfg.stereo, device=str(dev))
    video.images = None                                       # (uint8 full-resolution frames: not on the path)
    video.poses[:N] = d(g["poses"]); video.disps[:N] = d(g["disps"]); video.intrinsics[:N] = d(g["intrinsics"])
    video.disps_sens[:N] = d(g["disps_sens"])
    video.fmaps[:N] = d(g["fmaps"]); video.nets[:N] = d(g["nets"]); video.inps[:N] = d(g["inps"])
    if "disps_conf" in g:
        video.set_depth_confidence(slice(0, N), d(g["disps_conf"]))       # BASELINE configs[4]: per-pixel depth-confidence weights
    video.counter.value = N
    return video


def _sorted_edges(g):
    """edges grouped by source frame (the permutation of the seeded list that main() uses too)"""
    order = np.argsort(g["ii"], kind="stable")
    return order, g["ii"][order], g["jj"][order]


def _timed_updates(graph, video, g, poses0, disps0, steps, barrier):
    """`steps` graph.update iterations at the global-BA damping from the same poses / depths each -> (elapsed s, lookup ms, operator ms, ba ms)"""
    N = g["n_frames"]

    def step():
        video.poses[:N] = poses0; video.disps[:N] = disps0
        graph.update(1, N, itrs=g["itrs"], use_inactive=False, lm=g["lm"], ep=g["ep"])
    for _ in range(2):
        step()
    barrier()
    graph.phase_events = []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    ev, graph.phase_events = graph.phase_events, None
    mean = lambda k: float(np.mean([e[k].elapsed_time(e[k + 1]) for e in ev])) if ev else 0.0
    return elapsed, mean(0), mean(1), mean(2)


def product_class_line(args, g, dev, upd, conf, _unused):
    """N = 1: the step through droid_amd.factor_graph.FactorGraph.update on the whole graph, wall clock -- once as the class runs it
    (the gates' per-frame context term kept between keyframe insertions) and once recomputing it per iteration like the timed steps of
    `value` do.  HIP events inside update() (FactorGraph.phase_events) split it like ms_corr_lookup / ms_update_operator / ms_per_global_ba."""
    from droid_amd import synthetic as syn
    from droid_amd.factor_graph import FactorGraph
    cfg = syn.CONFIGS[args.config]
    N = g["n_frames"]
    d = lambda a, **kw: torch.as_tensor(np.ascontiguousarray(a), **kw).to(dev)
    video = _video_for(g, cfg, dev)
    _, ii, jj = _sorted_edges(g)                               # (g's edge arrays were permuted by main() already: identity here)
    graph = FactorGraph(video, upd, corr_impl="volume", max_factors=-1, upsample=False)
    graph.compute_upmask = True                                # like the timed steps and the reference: every head of the operator in every iteration
    graph.add_factors(d(ii), d(jj))
    poses0, disps0 = d(g["poses"]), d(g["disps"])
    sync = torch.cuda.synchronize
    res = {}
    for key, cache in (("ms_per_step", True), ("ms_per_step_recomputing_context", False)):
        graph.cache_context = cache
        for _ in range(max(0, args.warmup - 2)):
            _timed_updates(graph, video, g, poses0, disps0, 1, sync)
        el, lk, up, ba = _timed_updates(graph, video, g, poses0, disps0, args.steps, sync)
        res[key] = 1e3 * el / max(1, args.steps)
        if cache:
            res.update({"ms_corr_lookup": lk, "ms_update_operator": up, "ms_per_global_ba": ba})
    # what the class does by default for a graph built with upsample=False (the reference's callers' default): GraphAgg's upmask head, whose
    # only reader is DepthVideo.upsample, is not computed -- reported beside the figures above, never instead of them
    graph.cache_context, graph.compute_upmask = True, None
    el, _, up, _ = _timed_updates(graph, video, g, poses0, disps0, args.steps, sync)
    res["ms_per_step_without_upmask_head"] = 1e3 * el / max(1, args.steps)
    res["ms_update_operator_without_upmask_head"] = up
    res["steps"] = args.steps
    res["note"] = ("droid_amd.factor_graph.FactorGraph.update(1, N, itrs, lm, ep) on all edges, wall clock incl. its host logic; compare "
                   "ms_per_step with steady_state_cached_context.ms_per_step and ms_per_step_recomputing_context with the headline ms_per_step")
    graph.clear_edges()
    return res


def main_dist(args, dev, world, rank, local):
    """N > 1 (and DH_BENCH_DIST1=1: one rank on RCCL): the step through the PRODUCT class droid_amd.dist_graph.DistFactorGraph -- edges
    sharded by source frame, per-rank pyramid / hidden state / context table, DistBA's packed exchange; a step = DistFactorGraph.update
    at the global-BA damping from the same poses / depths.  Like the N = 1 steps it recomputes the gates' context term in every
    iteration (cache_context = False).  `value` = all ranks' edge-pixels / the slowest rank's time."""
    import droid_backends as db
    from droid_amd import synthetic as syn
    from droid_amd.dist_ba import DistBA
    from droid_amd.dist_graph import DistFactorGraph
    from droid_amd.update import UpdateModule, empty_state_dict
    from droid_amd.weights import deterministic_state_dict
    cfg = syn.CONFIGS[args.config]
    g = syn.make_graph(cfg, with_features=True)
    N, ht, wd = g["n_frames"], g["ht"], g["wd"]
    HW = ht * wd
    order, ii_all, jj_all = _sorted_edges(g)
    E_all = len(ii_all)
    d = lambda a, **kw: torch.as_tensor(np.ascontiguousarray(a), **kw).to(dev)

    class _SD:
        def state_dict(self):
            return empty_state_dict()
    upd = UpdateModule(dev).load_state_dict(deterministic_state_dict(_SD(), seed=1234))
    video = _video_for(g, cfg, dev)
    solver = DistBA(world, always_reduce=(world == 1))
    graph = DistFactorGraph(video, upd, corr_impl="volume", max_factors=-1, upsample=False, world=world, rank=rank, solver=solver)
    graph.compute_upmask = True                                # every head of the operator in every iteration, like the N = 1 steps
    graph.cache_context = False
    torch.cuda.synchronize()
    e0, e1 = _ev(), _ev()
    e0.record()
    graph.add_factors(d(ii_all), d(jj_all))                    # ownership ranges + THIS rank's pyramid records / hidden states
    e1.record(); torch.cuda.synchronize()
    ms_add = e0.elapsed_time(e1)
    mine = graph.local_index()
    E = int(mine.numel())
    # the synthetic targets of the graph (ground-truth reprojection + noise) as the previous targets of this rank's edges, like main()
    graph.target = d(g["targets"][order])[mine].permute(0, 2, 3, 1).contiguous()[None]
    poses0, disps0 = d(g["poses"]), d(g["disps"])

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        _timed_updates(graph, video, g, poses0, disps0, 0, barrier)
    elapsed, lk, up, ba = _timed_updates(graph, video, g, poses0, disps0, args.steps, barrier)
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # every rank ends an iteration with the same poses / depths
    chk = torch.cat([video.poses[:N].flatten(), video.disps[:N].flatten()]).double()
    lo_, hi_ = chk.clone(), chk.clone()
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN); dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    spread = float((hi_ - lo_).abs().max())
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {"rank": rank, "device": torch.cuda.get_device_name(local), "local_rank": local, "edges": E,
                                      "frames": [int(graph.frame_lo), int(min(graph.frame_hi, N))], "ms_corr_lookup": lk,
                                      "ms_update_operator": up, "ms_per_global_ba": ba, "pyramid_GB": graph.corr.bytes() / 1e9})

    def _ar_ms(x):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter(); dist.all_reduce(x); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        return sorted(ts)[2]
    ar_sys = _ar_ms(torch.zeros(max(1, int(solver.last_exchange_bytes) // 8), dtype=torch.float64, device=dev))
    ar_disps = _ar_ms(torch.zeros_like(video.disps))
    if rank == 0:
        ms = 1e3 * elapsed / max(1, args.steps)
        fused = bool(db.get_option("lookup_fused"))
        bpe = LOOKUP_FUSED_BYTES_PER_EP_F16 if fused else LOOKUP_BYTES_PER_EP_F16
        achieved = bpe * E * HW / (lk * 1e-3) / 1e9 if lk > 0 else 0.0
        out = {"metric": "BA update iterations/sec (edges*pixels/s), 512-KF graph", "value": E_all * HW / (ms * 1e-3), "unit": "edge-pixels/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32 (BA; fp64 solve) / f16 (correlation pyramid)", "data": "synthetic",
               "config": {"workload": "%s: %d keyframes, %d edges, %dx%d, ba itrs=%d lm=%g ep=%g%s" % (
                   cfg.name, N, E_all, ht, wd, g["itrs"], g["lm"], g["ep"],
                   ", stereo + sensor depth with per-pixel confidence weights (seeded, non-constant)" if cfg.stereo else ""),
                   "stages": "DistFactorGraph.update: reproject + corr lookup (4 levels, materialised fp16 pyramid%s) + ConvGRU update operator "
                             "(random-init weights) + edge-sharded ba (DistBA)" % (", fused with the correlation encoder's first layer" if fused else ""),
                   "parallelism": "edge-sharded x%d (droid_amd.dist_graph.DistFactorGraph)" % world},
               "ms_per_global_ba": ba, "ms_corr_lookup": lk, "ms_update_operator": up, "ms_add_factors_incl_pyramid_build": ms_add,
               "roofline": {"kernel": "pyr_lookup_corr0_kernel<64> on rank 0's %d edges (744 B/ep)" % E if fused else
                            "pyr_lookup_kernel<64> on rank 0's %d edges (880 B/ep)" % E, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                            "traffic_source": "measured at N = 1 only (the default line's in-run rocprofv3 --pmc pass)"},
               "dist": {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": per_rank,
                        "allreduce_bytes_per_gn_iteration": int(solver.last_exchange_bytes), "packed_exchange": bool(solver.last_exchange_packed),
                        "ms_allreduce_system": ar_sys, "ms_allreduce_disps": ar_disps, "ms_collectives_per_global_ba": g["itrs"] * ar_sys + ar_disps,
                        "max_state_spread_between_ranks": spread, "product_class": "droid_amd.dist_graph.DistFactorGraph"}}
        _emit(out)
    dist.destroy_process_group()


def main_lowmem(args, dev, world, rank):
    """`--lowmem`: the global-BA iteration (FactorGraph.update_lowmem, reference factor_graph.py:266-330) on the whole graph:
    per step reproject -> correlation features of every edge -> update operator -> ONE ba (lm = 1e-5, ep = 1e-2) over all
    keyframes.  `--lowmem-corr alt`: on-the-fly alt-correlation in chunks of --chunk-frames source frames (the reference's
    scheme; altcorr_mfma_kernel); `pyramid`: the pyramid of all edges built once per update_lowmem call (its build time is
    inside the timed region, amortised over the call's steps)."""
    assert world == 1, "--lowmem is a single-GPU measurement"
    import droid_backends as db
    from droid_amd import synthetic as syn
    from droid_amd import corr as corr_mod
    from droid_amd.depth_video import DepthVideo
    from droid_amd.factor_graph import FactorGraph
    from droid_amd.update import UpdateModule, empty_state_dict
    from droid_amd.weights import deterministic_state_dict
    cfg = syn.CONFIGS[args.config]
    g = syn.make_graph(cfg, with_features=True)
    N, ht, wd = g["n_frames"], g["ht"], g["wd"]
    HW = ht * wd
    E = len(g["ii"])
    d = lambda a, **kw: torch.as_tensor(np.ascontiguousarray(a), **kw).to(dev)

    class _SD:
        def state_dict(self):
            return empty_state_dict()
    upd = UpdateModule(dev).load_state_dict(deterministic_state_dict(_SD(), seed=1234))
    video = DepthVideo(image_size=[8 * ht, 8 * wd], buffer=N + 2, stereo=cfg.stereo, device=str(dev))
    poses0, disps0 = d(g["poses"]), d(g["disps"])
    video.poses[:N] = poses0; video.disps[:N] = disps0; video.intrinsics[:N] = d(g["intrinsics"])
    video.disps_sens[:N] = d(g["disps_sens"])
    video.fmaps[:N] = d(g["fmaps"]); video.nets[:N] = d(g["nets"]); video.inps[:N] = d(g["inps"])
    video.counter.value = N
    fg = FactorGraph(video, upd, corr_impl="alt", max_factors=16 * N, upsample=False, chunk_frames=args.chunk_frames)
    order = np.argsort(g["ii"], kind="stable")
    fg.add_factors(d(g["ii"][order]), d(g["jj"][order]))
    assert len(fg.ii) == E
    net0, target0, weight0 = fg._net.clone(), fg.target.clone(), fg.weight.clone()
    mode = args.lowmem_corr
    if mode == "auto":
        mode = "pyramid" if fg._pyramid_fits(E, ht, wd) else "alt"
    steps_per_call = 8                                       # update_lowmem's default (droid_backend.py calls it with 7 and 12)
    # HIP-event time of the alt-correlation launches (4 levels per chunk)
    alt_ev = []
    orig_call = corr_mod.AltCorrBlock.__call__

    def timed_call(self, *a, **kw):
        e0, e1 = _ev(), _ev()
        e0.record(); r = orig_call(self, *a, **kw); e1.record()
        alt_ev.append((e0, e1))
        return r
    corr_mod.AltCorrBlock.__call__ = timed_call

    def call():
        video.poses[:N] = poses0; video.disps[:N] = disps0
        fg._net.copy_(net0); fg._glo = None; fg.target = target0.clone(); fg.weight = weight0.clone(); fg.damping.fill_(1e-6)
        fg.update_lowmem(steps=steps_per_call, corr=mode)

    call(); torch.cuda.synchronize()                         # code objects, LDS opt-ins, allocator pool
    for _ in range(max(0, args.warmup - 1)):
        call()
    torch.cuda.synchronize()
    alt_ev.clear()
    ncalls = max(1, args.steps // steps_per_call)
    t0 = time.perf_counter()
    for _ in range(ncalls):
        call()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    nsteps = ncalls * steps_per_call
    ms = 1e3 * elapsed / nsteps
    alt_ms = sum(a.elapsed_time(b) for a, b in alt_ev) / nsteps if alt_ev else None
    out = {"metric": "global-BA update iterations/sec (edges*pixels/s), FactorGraph.update_lowmem", "value": E * HW / (ms * 1e-3),
           "unit": "edge-pixels/s", "n_gpus": 1, "steps": nsteps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32 (BA; fp64 solve) / f16 (features)", "data": "synthetic",
           "config": {"workload": "%s: %d keyframes, %d edges, %dx%d, update_lowmem(steps=%d) calls, ba itrs=2 lm=1e-5 ep=1e-2" % (
               cfg.name, N, E, ht, wd, steps_per_call), "correlation": mode, "chunk_frames": args.chunk_frames}}
    if alt_ms is not None:
        flops = 65536.0 * E * HW                              # 4 levels x 64 taps x 128 channels x 2
        tf = flops / (alt_ms * 1e-3) / 1e12
        out["roofline_altcorr"] = {"kernel": "altcorr_mfma_kernel (4 launches per chunk; HIP events around AltCorrBlock.__call__, "
                                   "incl. its level stack copies)", "bound": "mfma", "ms_per_step": alt_ms, "achieved": tf, "peak": 2500.0,
                                   "unit": "TFLOP/s", "frac": tf / 2500.0, "algorithmic_flops": flops}
    _emit(out)


if __name__ == "__main__":
    main()
