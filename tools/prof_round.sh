set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_serial -o serial -- python3 $R/bench.py --no-cpu-baseline --no-alts --steps 100 --warmup 10 > $R/gpurun_out/prof_serial.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_default -o default -- python3 $R/bench.py --graph --no-cpu-baseline --no-alts --steps 100 --warmup 10 > $R/gpurun_out/prof_default.log 2>&1
cd $R
python bench.py > gpurun_out/bench_final.log 2>&1
tail -1 gpurun_out/bench_final.log
python tools/blas_yardstick.py > gpurun_out/yard.log 2>&1
python tools/deep_bench.py > gpurun_out/deep_bench.log 2>&1
