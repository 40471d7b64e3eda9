#!/usr/bin/env python
"""HBM rates of the three access mixes the path's kernels have, on large tensors (torch's own streaming kernels, HIP events, median of 7):
read-only (sum), write-only (fill_ / zero_), copy (read + write).  The write-heavy kernels of the path -- the flow encoder's stem (3.2 GB of
activations written per launch, 0.1 GB read), the upmask head, the correlation-pyramid build (105 GB of records per 4096 edges) -- are to be
judged against the WRITE rate, not against the 8 TB/s of the spec or the copy rate.
    python scripts/ubench/hbm_rw.py [--gb 8]"""
import argparse
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--gb", type=float, default=8.0)
a = ap.parse_args()
n = int(a.gb * (1 << 30)) // 4
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
x.normal_(); y.zero_()


def timed(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


gb = n * 4 / 1e9
xh = x.view(torch.float16)
rows = [("read-only   (torch.sum, fp32)", lambda: x.sum(), gb),
        ("read-only   (torch.sum of fp16 as int16 view)", lambda: xh.view(torch.int16).sum(), gb),
        ("write-only  (fill_(1.5), fp32)", lambda: y.fill_(1.5), gb),
        ("write-only  (zero_)", lambda: y.zero_(), gb),
        ("write-only  (fill_ fp16 view)", lambda: y.view(torch.float16).fill_(0.5), gb),
        ("copy        (y.copy_(x): read + write)", lambda: y.copy_(x), 2 * gb),
        ("read + small write (x[: n // 32] = relu: 1/32)", lambda: torch.relu_(x[: n // 32]), 2 * gb / 32)]
for name, fn, moved in rows:
    ms = timed(fn)
    print("%-50s %8.3f ms  %7.2f TB/s of bytes moved" % (name, ms, moved / (ms * 1e-3) / 1e3), flush=True)
