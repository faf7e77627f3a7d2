#!/usr/bin/env python3
"""Median duration of every kernel of one training step, in issue order, from a rocprofv3 kernel trace CSV
(tools/r06_prof_shape.sh): python tools/trace_steps.py path/to/*_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first = "k_cast_pad_bf16"
idx = [i for i, n in enumerate(names) if first in n]
seq = collections.defaultdict(list)
n_use = min(60, len(idx) - 1)
for a, b in zip(idx[-n_use - 1:-1], idx[-n_use:]):
    for j, r in enumerate(rows[a:b]):
        seq[(j, r["Kernel_Name"][:100])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    seq[(99, "step (start to start)")].append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
tot = 0.0
for k in sorted(seq):
    v = sorted(seq[k])
    if k[0] != 99:
        tot += v[len(v) // 2]
    print("%2d %8.1f us  %s" % (k[0], v[len(v) // 2], k[1]))
print("sum of kernel medians %.1f us" % tot)
