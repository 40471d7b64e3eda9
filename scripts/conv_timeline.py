#!/usr/bin/env python
"""Per-CU phase timeline of the gate convolution ("record the per-phase timeline that shows where the overlap
fails").  -DDH_ABLATION build only: every workgroup of conv3x3_halo2_kernel (256-pixel tile, TWO workgroups per CU: the product)
and conv3x3_halo3_kernel (512-pixel tile, ONE workgroup per CU: 13 % fewer joules, no faster) stores the 100 MHz wall clock at
kernel entry / first fetches issued / first barrier passed / main loop left / epilogue done plus the CU it ran on
(dh_conv_set_timestamps).  Per CU the intervals [first barrier, main loop left] are the time its matrix pipe has work; everything
else -- the accumulator start values from the context term, the first halo + weight group in flight, the GRU epilogue and its
staged store -- is exposed unless another workgroup of the same CU is in ITS main loop.

    python scripts/conv_timeline.py [--edges 1024] [--out gpurun_out/conv_timeline.json]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd", "ablation"), os.path.join(ROOT, "droid-slam_amd")]
import numpy as np
import torch
import droid_backends as db
assert db.get_option("ablation_build") == 1, "needs the -DDH_ABLATION build (DROID_HIP_ABLATION=1 python droid-slam_amd/build.py)"
from droid_amd.update import UpdateModule, EPI_RELU, EPI_GRU_ZR, EPI_GRU_Q, EPI_HEADS0
from droid_amd.weights import deterministic_state_dict
from oracle import update as oupd          # (shape template of the state dict only)

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=1024)
ap.add_argument("--out", default=None)
ap.add_argument("--only", default=None, help="substring filter on the case names")
ap.add_argument("--gate64", action="store_true", help="round 6: the product form against the SAME launches on 64-cout tiles at three workgroups "
                "per CU (conv3x3_halo64_kernel, option conv_gate64) instead of against the 512-pixel form")
a = ap.parse_args()
E, K, h, w = a.edges, max(1, a.edges // 8), 48, 64


class _SD:
    def state_dict(self):
        return oupd.empty_state_dict()


torch.manual_seed(0)
upd = UpdateModule("cuda").load_state_dict(deterministic_state_dict(_SD(), seed=7))
P = upd.params
net = torch.tanh(torch.randn(E, h, w, 128, device="cuda")).half()
c = torch.relu(torch.randn(E, h, w, 128, device="cuda")).half()
f = torch.relu(torch.randn(E, h, w, 64, device="cuda")).half()
inp_frames = torch.relu(torch.randn(K, h, w, 128, device="cuda")).half()
idx = (torch.arange(E, device="cuda") // 8).clamp(max=K - 1)
ctx_pm = upd.context_term(inp_frames, tiled=False)          # pixel-major [K,h,w,384] (rounds 2-4)
ctx_tl = upd.context_term(inp_frames, tiled=True)           # accumulator-tile layout (round 5 default)
gzr = torch.randn(E, 256, device="cuda") * 0.1
zr = torch.empty(E, h, w, 256, device="cuda", dtype=torch.float16)
nwg = (E * h * w // 256) * (4 if a.gate64 else 2)
buf = torch.zeros(nwg + 2048, 8, dtype=torch.int64, device="cuda")          # + 256 workgroups x 64 per-step stamps


def analyse(t, n):
    """t [n, 8] int64 (10 ns ticks) -> statistics over the steady part of the launch"""
    t = t[:n].astype(np.int64)
    cu = (t[:, 5] >> 8 & 0xFF) | ((t[:, 6] & 0xF) << 8)                     # (se, sh, cu) of HW_ID + XCC_ID
    t0 = t[:, 0].min()
    ph = (t[:, :5] - t0) * 0.01                                             # microseconds
    span = ph[:, 4].max()
    out = {"workgroups": int(n), "compute_units_seen": int(len(np.unique(cu))), "launch_us": float(span),
           "prologue_us_mean": float((ph[:, 2] - ph[:, 0]).mean()), "prologue_us_p90": float(np.quantile(ph[:, 2] - ph[:, 0], 0.9)),
           "of_which_until_fetches_issued_us": float((ph[:, 1] - ph[:, 0]).mean()),
           "main_loop_us_mean": float((ph[:, 3] - ph[:, 2]).mean()),
           "epilogue_us_mean": float((ph[:, 4] - ph[:, 3]).mean()), "epilogue_us_p90": float(np.quantile(ph[:, 4] - ph[:, 3], 0.9))}
    # per CU: fraction of its busy span during which at least one resident workgroup is inside its main loop, and the mean number
    # of workgroups resident / in their main loop
    cover, resident, inloop = [], [], []
    for u in np.unique(cu):
        m = cu == u
        if m.sum() < 4:
            continue
        s0, s1 = ph[m, 0].min(), ph[m, 4].max()
        ev = sorted([(x, 1) for x in ph[m, 2]] + [(x, -1) for x in ph[m, 3]])
        depth, last, on = 0, s0, 0.0
        for x, d in ev:
            if depth > 0:
                on += x - last
            depth += d; last = x
        cover.append(on / (s1 - s0))
        resident.append((ph[m, 4] - ph[m, 0]).sum() / (s1 - s0)); inloop.append((ph[m, 3] - ph[m, 2]).sum() / (s1 - s0))
    out.update({"cu_main_loop_coverage_mean": float(np.mean(cover)), "cu_main_loop_coverage_min": float(np.min(cover)),
                "workgroups_resident_per_cu_mean": float(np.mean(resident)), "workgroups_in_main_loop_per_cu_mean": float(np.mean(inloop))})
    return out


res = {"edges": E, "what": __doc__.split("\n\n")[0], "cases": []}
for variant in (0, 1):
    halo3 = variant if not a.gate64 else 0
    gate64 = variant if a.gate64 else 0
    db.set_option("conv_halo3", halo3)
    db.set_option("conv_gate64", 3 if gate64 else 0)
    upd = UpdateModule("cuda").load_state_dict(deterministic_state_dict(_SD(), seed=7)); P = upd.params       # (weights packed under the option)
    cases = {"relu (bare convolution, plain staged epilogue)": lambda: P["zr_e"]([net, c, f], EPI_RELU, out=zr),
             "gru (GRU epilogue, accumulators from zero)": lambda: P["zr_e"]([net, c, f], EPI_GRU_ZR, out=zr, gterm=gzr, aux0=net),
             "gru+cinit, pixel-major context term (rounds 2-4)": lambda: P["zr_e"]([net, c, f], EPI_GRU_ZR, out=zr, gterm=gzr, aux0=net, cinit=ctx_pm, cinit_idx=idx, cinit_off=0)}
    if not halo3:             # (the 512-pixel form reads pixel-major start values only)
        cases["gru+cinit, accumulator-tile context term (the product's z|r launch)"] = lambda: P["zr_e"]([net, c, f], EPI_GRU_ZR, out=zr, gterm=gzr, aux0=net, cinit=ctx_tl, cinit_idx=idx, cinit_off=0)
    if not halo3:             # the q gate (one cout tile per pixel tile; its epilogue reads z and the old state)
        gq = torch.randn(E, 128, device="cuda") * 0.1
        zr_in = torch.rand(E, h, w, 256, device="cuda").half()
        outq = torch.empty(E, h, w, 128, device="cuda", dtype=torch.float16)
        cases["q: relu (bare convolution 320 -> 128)"] = lambda: P["q_e"]([zr_in[..., 128:], c, f], EPI_RELU, out=outq)
        cases["q: gru (tanh + state update, accumulators from zero)"] = lambda: P["q_e"]([zr_in[..., 128:], c, f], EPI_GRU_Q, out=outq, gterm=gq, aux0=net, aux1=zr_in)
        cases["q: gru+cinit, accumulator-tile context term (the product's q launch)"] = lambda: P["q_e"]([zr_in[..., 128:], c, f], EPI_GRU_Q, out=outq, gterm=gq, aux0=net, aux1=zr_in, cinit=ctx_tl, cinit_idx=idx, cinit_off=256)
    if not halo3 and not gate64:             # the flow encoder's 7x7 stem (conv7x7_c4_kernel: stamps 1 = staging issued, 2 = staged, 3 = MFMAs done)
        flow8 = torch.zeros(E, h, w, 8, device="cuda", dtype=torch.float16); flow8[..., :4] = torch.randn(E, h, w, 4, device="cuda").half()
        xs = torch.empty(E, h, w, 128, device="cuda", dtype=torch.float16)
        cases["stem: 7x7 on 4 channels -> 128 (flow_encoder.0)"] = lambda: P["flow0"]([flow8], EPI_RELU, out=xs)
    if gate64:                # (launches that stay in conv3x3_halo2_kernel under the option are not repeated)
        cases = {k: v for k, v in cases.items() if "pixel-major" not in k}
        x128 = torch.empty(E, h, w, 128, device="cuda", dtype=torch.float16)
        cases["128 -> 128 relu (corr_encoder.2 / agg.conv1)"] = lambda: P["agg1"]([net], EPI_RELU, out=x128)
    if not halo3 and not gate64:             # the other 3x3 launches of the operator that run in this kernel
        part = torch.empty(2, E * h // 4, 6, 64, 4, dtype=torch.float32, device="cuda")
        w2p = P["heads2_fused"][0]
        x128 = torch.empty(E, h, w, 128, device="cuda", dtype=torch.float16)
        cases["128 -> 128 relu (corr_encoder.2 / agg.conv1)"] = lambda: P["agg1"]([net], EPI_RELU, out=x128)
        cases["heads: 128 -> 256 + the fused second layer (EPI_HEADS0)"] = lambda: P["heads0"]([net], EPI_HEADS0, aux1=w2p, red=part)
        cases["context term 128 -> 384, accumulator-tile output (512 frames at E = 4096)"] = lambda: upd.context_term(inp_frames, tiled=True)
    tiles = E * h * w // 256
    nwg_of = lambda name: (tiles if name.startswith(("q:", "128 -> 128", "stem:")) else (K * h * w // 256) * 3 if name.startswith("context term") else 2 * tiles) * (2 if gate64 else 1) // (2 if halo3 else 1)
    for name, fn in cases.items():
        if a.only and a.only not in name:
            continue
        db.conv_set_timestamps(None)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms_plain = e0.elapsed_time(e1)
        buf.zero_(); db.conv_set_timestamps(buf)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms_ts = e0.elapsed_time(e1)
        db.conv_set_timestamps(None)
        tb = buf.cpu().numpy()
        n = nwg_of(name)                                     # workgroups of this launch (rows 0 .. n-1; the per-step stamps follow)
        assert (tb[:n, 4] != 0).all() and (n >= nwg or (tb[n + 2048:, 4] == 0).all()), name
        r = analyse(tb, n)
        if not halo3:                                        # per-step stamps of 256 workgroups from the middle of the launch
            st = tb[n:n + 2048].reshape(256, 64).astype(np.int64)
            ns = int((st[0] != 0).sum())
            if ns > 2:
                d = np.diff(st[:, :ns], axis=1) * 0.01       # [256, ns - 1] microseconds per step
                r["steps"] = ns
                r["step_us_median_by_index"] = [float(x) for x in np.median(d, axis=0)]
                r["step_us_mean"] = float(d.mean()); r["step_us_p90"] = float(np.quantile(d, 0.9)); r["step_us_min"] = float(d.min())
        r.update({"kernel": "conv3x3_halo64_kernel (256 px x 64 couts, three workgroups per CU)" if gate64 else "conv3x3_halo3_kernel (512-px tile, one workgroup per CU)" if halo3 else "conv3x3_halo2_kernel (256-px tile, two workgroups per CU)",
                  "launch": name, "ms_without_timestamps": ms_plain, "ms_with_timestamps": ms_ts})
        res["cases"].append(r)
        print("%-58s %-70s %.3f ms (%.3f with timestamps): prologue %.1f us (p90 %.1f), main loop %.1f us, epilogue %.1f us (p90 %.1f); per CU: "
              "%.2f workgroups resident, %.2f in their main loop, main-loop coverage %.3f (min %.3f)" % (
                  r["kernel"], name, ms_plain, ms_ts, r["prologue_us_mean"], r["prologue_us_p90"], r["main_loop_us_mean"], r["epilogue_us_mean"],
                  r["epilogue_us_p90"], r["workgroups_resident_per_cu_mean"], r["workgroups_in_main_loop_per_cu_mean"],
                  r["cu_main_loop_coverage_mean"], r["cu_main_loop_coverage_min"]), flush=True)
        if "steps" in r:
            print("      per step (us, median over 256 workgroups, step 0 -> 1 first): %s | mean %.2f p90 %.2f min %.2f" % (
                " ".join("%.1f" % x for x in r["step_us_median_by_index"]), r["step_us_mean"], r["step_us_p90"], r["step_us_min"]), flush=True)
db.set_option("conv_halo3", 0); db.set_option("conv_gate64", 0)
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
