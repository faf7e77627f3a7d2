# Round-2 evidence run (one gpurun call): kernel-trace stats of the default bench command, FETCH_SIZE / WRITE_SIZE PMC
# passes (separate, as MI355X_MICROARCH.md 'HBM' prescribes), per-kernel timeline, then the plain bench lines.
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-alts --steps 100 --warmup 10 --repeats 5 > $O/r02_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r02_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-alts --steps 20 --warmup 3 --repeats 1 > $O/r02_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r02_write -o w -- python3 $R/bench.py --no-cpu-baseline --no-alts --steps 20 --warmup 3 --repeats 1 > $O/r02_write.log 2>&1
cd $R
python tools/timeline_csv.py $(find $O/r02_stats -name "*kernel_trace.csv" | head -1) 5 > $O/r02_timeline.txt
python bench.py > $O/r02_bench.json 2>$O/r02_bench.err
tail -1 $O/r02_bench.json | cut -c1-300
