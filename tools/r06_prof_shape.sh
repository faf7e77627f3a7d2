#!/bin/bash
# rocprofv3 kernel stats of the training step at one shape: tools/r06_prof_shape.sh TAG S H L B
set -e
export TMPDIR=/tmp
TAG=$1; shift
R=$PWD; O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python3 $R/tools/step_time.py --shape "$@" --steps 100 --reps 1 > $O/prof.log 2>&1
cd $R
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python3 tools/trace_steps.py $(find $O/prof -name "*kernel_trace.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:14]:
    n = r["Name"][:110]
    print("%9.1f us x %5s  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], n))
PY
