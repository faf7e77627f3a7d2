#!/bin/bash
# rocprofv3 kernel stats of the deep variant's step (eager launches): tools/r06_prof_deep.sh TAG
set -e
export TMPDIR=/tmp
TAG=${1:-r06_deep}
R=$PWD; O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python3 $R/tools/deep_bench.py --no-graph --steps 60 > $O/prof.log 2>&1
cd $R
python3 - $(find $O/prof -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%9.1f us x %5s = %8.1f us/step  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["TotalDurationNs"]) / 1e3 / 81, r["Name"][:120]))
PY
