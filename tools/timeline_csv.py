#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: start offset, duration and queue of every
kernel of the N-th step from the end (a step starts at k_cast_pad_bf16).  python tools/timeline_csv.py trace.csv [n_back]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_cast_pad_bf16" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 100000]
i0, i1 = starts[-back - 1], starts[-back]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    nm = r["Kernel_Name"].replace("void rv::", "").replace("(anonymous namespace)::", "")[:70]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("q%-3s %8.1f -> %8.1f  (%6.1f us)  %s" % (r["Queue_Id"], s / 1e3, e / 1e3, (e - s) / 1e3, nm))
print("step span %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
