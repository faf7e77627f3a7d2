# Profile run (one gpurun call): kernel-trace stats + per-kernel timeline of the default bench command.
# usage: bash tools/prof_step.sh TAG [extra bench.py flags]
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-alts --step-kernels-only --steps 100 --warmup 10 --repeats 5 "$@" > $O/${TAG}_stats.log 2>&1
cd $R
python tools/timeline_csv.py $(find $O/${TAG}_stats -name "*kernel_trace.csv" | head -1) 5 > $O/${TAG}_timeline.txt
python tools/prof_summary.py $O/${TAG}_stats 511 > $O/${TAG}_kernel_summary.txt
cp $(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
rm -rf $O/${TAG}_stats
cat $O/${TAG}_timeline.txt
