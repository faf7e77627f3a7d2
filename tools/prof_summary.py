#!/usr/bin/env python3
"""Print a per-kernel table from a rocprofv3 --kernel-trace --stats CSV directory."""
import csv
import glob
import sys


def main(d, steps=None):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not f:
        print("no kernel_stats.csv under", d)
        return
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows:
        n = r["Name"].replace("void rv::", "").replace("rv::", "").replace("(anonymous namespace)::", "")[:90]
        print("%-90s calls %5s avg %8.1f us min %8.1f  %5.1f%%" % (
            n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            100 * float(r["TotalDurationNs"]) / tot))
    if steps:
        print("total kernel time per step: %.1f us" % (tot / 1e3 / float(steps)))


if __name__ == "__main__":
    main(*sys.argv[1:])
