# Kernel-trace stats + one step's timeline of the fp8 weight path (one gpurun call).  usage: bash tools/prof_fp8.sh TAG
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=$1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_fp8_stats -o s -- python3 $R/tools/fp8_steps.py 510 > $O/${TAG}_fp8_stats.log 2>&1
cd $R
python tools/timeline_csv.py $(find $O/${TAG}_fp8_stats -name "*kernel_trace.csv" | head -1) 5 > $O/${TAG}_fp8_timeline.txt
python tools/prof_summary.py $O/${TAG}_fp8_stats 510 > $O/${TAG}_fp8_kernel_summary.txt
rm -rf $O/${TAG}_fp8_stats
cat $O/${TAG}_fp8_timeline.txt; head -14 $O/${TAG}_fp8_kernel_summary.txt
