#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace as a per-kernel table (calls, total/avg/min/max ns, %)."""
import sqlite3, sys
def main(path, top=40):
    con = sqlite3.connect(path); cur = con.cursor()
    q = """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("%-90s %8s %14s %12s %12s %12s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%"))
    for r in rows[:top]:
        print("%-90s %8d %14d %12.0f %12d %12d %6.2f" % (r[0][:90], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    print("TOTAL kernel time ns:", tot)
if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
