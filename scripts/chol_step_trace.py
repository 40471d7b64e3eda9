#!/usr/bin/env python
"""Per-launch durations of the Cholesky look-ahead steps of one global BA (C3), from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o run -- python scripts/chol_step_trace.py run
    python scripts/chol_step_trace.py parse OUT/**/run_kernel_trace.csv"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
if sys.argv[1] == "run":
    import numpy as np, torch
    import droid_backends as db
    from droid_amd import synthetic as syn
    g = syn.make_graph("C3")
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    args = [d(g[k]) for k in ("intrinsics", "disps_sens", "targets", "weights", "eta", "ii", "jj")]
    for _ in range(3):
        p, q = d(g["poses"]), d(g["disps"])
        db.ba(p, q, *args, 1, g["n_frames"], 2, g["lm"], g["ep"], False)
        torch.cuda.synchronize()
else:
    rows = [r for r in csv.DictReader(open(sys.argv[2]))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows if "chol_step_kernel" in r["Kernel_Name"]]
    n = 47
    last = steps[-2 * n:]                                  # the last ba call: two factorisations
    for it in range(2):
        s = last[it * n:(it + 1) * n]
        print("iteration %d: sum %.1f us; steps " % (it, sum(s)) + " ".join("%.0f" % v for v in s))
    ba = [r for r in rows if any(k in r["Kernel_Name"] for k in ("chol_", "ba_"))]
    t0, t1 = int(ba[-1]["End_Timestamp"]), None
    names = {}
    for r in rows[-400:]:
        if any(k in r["Kernel_Name"] for k in ("chol_", "ba_")):
            k = r["Kernel_Name"].split("(")[0].split("<")[0]
            names.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    print({k: (len(v), round(sum(v), 1)) for k, v in names.items()})
