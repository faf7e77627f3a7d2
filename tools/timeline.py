#!/usr/bin/env python3
"""Print the kernel timeline of the last full step from a rocprofv3 --kernel-trace CSV dir."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts with k_cast_pad on the x batch (grid large); find the last two such starts
starts = [i for i, r in enumerate(rows) if "k_cast_pad" in r["Kernel_Name"] and int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) > 100000]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]["Start_Timestamp"])
busy_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    n = r["Kernel_Name"].replace("void rv::", "").replace("rv::", "").replace("(anonymous namespace)::", "")[:62]
    gap = s - (busy_end - t0)
    print("%8.1f -> %8.1f  (%6.1f us) gap %6.1f  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, gap / 1e3, n))
    busy_end = max(busy_end, int(r["End_Timestamp"]))
print("step span: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
