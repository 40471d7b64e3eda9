#!/usr/bin/env python
"""Timeline of ONE update iteration of bench.py from a rocprofv3 kernel trace: which kernels run in what order, how long,
and how long the GPU idles between them (launch gaps, host-side stalls, copies).

    rocprofv3 --kernel-trace --output-format csv -d OUT -o run -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline \
        --no-sensitivity --no-pmc --no-lowmem --no-check
    python scripts/step_timeline.py OUT/**/run_kernel_trace.csv [--step -2]

A step is delimited by two consecutive launches of the fused lookup kernel (one per iteration)."""
import argparse
import csv
import glob

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--step", type=int, default=-2, help="which step (index into the list of lookup launches)")
ap.add_argument("--marker", default="pyr_lookup_corr0_kernel")
ap.add_argument("--gap-us", type=float, default=3.0)
a = ap.parse_args()
path = glob.glob(a.trace, recursive=True)[0]
rows = [r for r in csv.DictReader(open(path))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
idx = a.step if a.step >= 0 else len(marks) + a.step
assert 0 <= idx < len(marks) - 1, "need a following lookup launch to close the step"
i0, i1 = marks[idx], marks[idx + 1]
seg = rows[i0:i1]
t0 = int(seg[0]["Start_Timestamp"])
wall = (int(rows[i1]["Start_Timestamp"]) - t0) * 1e-3
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
busy, prev_end, gaps, agg = 0.0, t0, [], {}
print("%9s %9s %8s  kernel" % ("start us", "dur us", "gap us"))
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) * 1e-3
    d = (e - s) * 1e-3
    busy += d
    k = short(r["Kernel_Name"])
    agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += d
    if gap > a.gap_us:
        gaps.append((gap, k))
    print("%9.1f %9.1f %8.1f  %s" % ((s - t0) * 1e-3, d, gap, k))
    prev_end = max(prev_end, e)
tail = (int(rows[i1]["Start_Timestamp"]) - prev_end) * 1e-3
print("\nstep: %.1f us lookup-to-lookup, %d launches, kernels busy %.1f us, idle %.1f us (of which %.1f us before the next lookup)" % (
    wall, len(seg), busy, wall - busy, tail))
print("gaps > %.0f us: %d, together %.1f us" % (a.gap_us, len(gaps), sum(g for g, _ in gaps)))
for g, k in sorted(gaps, reverse=True)[:25]:
    print("   %8.1f us before %s" % (g, k))
print("\nby kernel:")
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("   %9.1f us  %4d x  %s" % (d, n, k))
