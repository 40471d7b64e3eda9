#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats --output-format csv -> the table format of profiles/*_kernel_stats.md.
usage: python scripts/kernel_stats_md.py <dir>/run_kernel_stats.csv [top-n]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return ("torch: " + n[:60]) if n.startswith("at::") or "at::native" in n else n[:90]


tot = sum(float(r["TotalDurationNs"]) for r in rows)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in rows[:top]:
    print("| %s | %d | %.3f | %.1f | %.2f |" % (short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                               float(r["AverageNs"]) / 1e3, 100.0 * float(r["TotalDurationNs"]) / tot))
print("\ntotal kernel time %.1f ms over %d dispatches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
