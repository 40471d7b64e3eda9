#!/usr/bin/env python
"""What the gate convolutions pay for their accumulator start values (the per-frame context term, fp32) and for the GRU epilogue:
the z|r and q launches of the update operator at C3 size (4096 edges, 512 source frames, 48x64) as
  relu        the bare convolution (320 -> 256 / 128) with the plain staged epilogue,
  gru         the GRU epilogue (sigmoid / r*net resp. tanh and the state update; operand loads from net / z|r), accumulators from zero,
  gru+cinit   the product launch: accumulators start from ctx[frame] (64 fp32 loads per lane and tile).
    python scripts/bench_gates.py [--edges 4096] [--reps 5]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT] + ([os.path.join(ROOT, "droid-slam_amd", os.environ["DH_LIB_DIR"])] if os.environ.get("DH_LIB_DIR") else []) + [os.path.join(ROOT, "droid-slam_amd")]
import torch
import droid_backends as db
from droid_amd.update import UpdateModule, EPI_RELU, EPI_GRU_ZR, EPI_GRU_Q
from droid_amd.weights import deterministic_state_dict
from oracle import update as oupd          # (shape template of the state dict only)

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=4096)
ap.add_argument("--reps", type=int, default=5)
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
ctx = upd.context_term(inp_frames, tiled=False)          # pixel-major start values (rounds 2-4)
ctx_t = upd.context_term(inp_frames, tiled=True)         # accumulator-tile layout (round 5 default)
gzr = torch.randn(E, 256, device="cuda") * 0.1
gq = torch.randn(E, 128, device="cuda") * 0.1
zr = torch.empty(E, h, w, 256, device="cuda", dtype=torch.float16)
out = torch.empty(E, h, w, 128, device="cuda", dtype=torch.float16)


def timed(fn):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


fl = lambda cout: 2.0 * E * h * w * 320 * 9 * cout
for rep in range(2):
    rows = [
        ("z|r relu", lambda: P["zr_e"]([net, c, f], EPI_RELU, out=zr), 256),
        ("z|r gru", lambda: P["zr_e"]([net, c, f], EPI_GRU_ZR, out=zr, gterm=gzr, aux0=net), 256),
        ("z|r gru+cinit", lambda: P["zr_e"]([net, c, f], EPI_GRU_ZR, out=zr, gterm=gzr, aux0=net, cinit=ctx, cinit_idx=idx, cinit_off=0), 256),
        ("z|r gru+cinit-t", lambda: P["zr_e"]([net, c, f], EPI_GRU_ZR, out=zr, gterm=gzr, aux0=net, cinit=ctx_t, cinit_idx=idx, cinit_off=0), 256),
        ("q relu", lambda: P["q_e"]([zr[..., 128:], c, f], EPI_RELU, out=out), 128),
        ("q gru", lambda: P["q_e"]([zr[..., 128:], c, f], EPI_GRU_Q, out=out, gterm=gq, aux0=net, aux1=zr), 128),
        ("q gru+cinit", lambda: P["q_e"]([zr[..., 128:], c, f], EPI_GRU_Q, out=out, gterm=gq, aux0=net, aux1=zr, cinit=ctx, cinit_idx=idx, cinit_off=256), 128),
        ("q gru+cinit-t", lambda: P["q_e"]([zr[..., 128:], c, f], EPI_GRU_Q, out=out, gterm=gq, aux0=net, aux1=zr, cinit=ctx_t, cinit_idx=idx, cinit_off=256), 128),
        ("ctx conv", lambda: upd.context_term(inp_frames, tiled=False), 0),
        ("ctx conv tiled", lambda: upd.context_term(inp_frames, tiled=True), 0),
    ]
    for name, fn, cout in rows:
        ms = timed(fn)
        print("%-16s %7.3f ms  %.3f PFLOP/s" % (name, ms, (fl(cout) if cout else 2.0 * K * h * w * 128 * 9 * 384) / ms / 1e12), flush=True)
