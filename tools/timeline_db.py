#!/usr/bin/env python3
"""Kernel timeline of one late step from a rocprofv3 rocpd .db (kernel-trace):
  python tools/timeline_db.py gpurun_out/prof/x_results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, stream_id from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "k_cast_pad" in r[0] and r[3] > 100000]
a, b = starts[-3], starts[-2]
t0 = rows[a][1]
busy_end = t0
for n, s, e, g, st in rows[a:b]:
    nm = n.replace("void rv::", "").replace("rv::", "").replace("(anonymous namespace)::", "")[:70]
    print("%8.1f -> %8.1f (%6.1f us) gap %6.1f s%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - busy_end) / 1e3, st, nm))
    busy_end = max(busy_end, e)
print("step span: %.1f us; kernel time in step: %.1f us" % ((rows[b][1] - t0) / 1e3, sum(r[2] - r[1] for r in rows[a:b]) / 1e3))
