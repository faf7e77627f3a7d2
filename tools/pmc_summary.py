#!/usr/bin/env python3
"""Average PMC counters per kernel (rocprofv3 --pmc CSV dir)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].replace("void rv::", "")[:58] + " g" + r["Grid_Size"]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, dd in agg.items():
    if filt not in k:
        continue
    print("%s  (%.1f us)" % (k, sum(dur[k]) / len(dur[k])))
    print("    " + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(dd.items())))
