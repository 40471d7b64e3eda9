#!/usr/bin/env python
"""Per-kernel summary (calls, total ms, average us, share) from a rocprofv3 sqlite trace (<name>_results.db), in the format
of profiles/*_kernel_stats.md.  usage: python scripts/rocpd_stats.py gpurun_out/x/prof/name_results.db [top-n]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rows = c.execute("select name, count(*), sum(end - start), avg(end - start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return ("torch: " + n[:60]) if n.startswith("at::") or "at::native" in n else n[:90]


print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for n, k, t, a in rows[:top]:
    print("| %s | %d | %.3f | %.1f | %.2f |" % (short(n), k, t / 1e6, a / 1e3, 100.0 * t / tot))
print("\ntotal kernel time %.1f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
