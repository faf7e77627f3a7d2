#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2 writes a rocpd sqlite .db) -> the same per-kernel stats CSV that
`--stats` prints: Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev.
  python tools/prof_db_summary.py gpurun_out/prof_serial/serial_results.db profiles/out.csv"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "avg(duration*duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for n, calls, s, a, mn, mx, a2 in rows:
            w.writerow([n, calls, s, round(a, 3), round(100.0 * s / tot, 3), mn, mx, round(max(a2 - a * a, 0) ** 0.5, 3)])
    for n, calls, s, a, mn, mx, a2 in rows[:16]:
        print("%-100s calls %5d avg %8.1f us min %8.1f  %5.1f%%" % (n.replace("void rv::", "").replace("(anonymous namespace)::", "")[:100], calls, a / 1e3, mn / 1e3, 100 * s / tot))


if __name__ == "__main__":
    main(*sys.argv[1:])
