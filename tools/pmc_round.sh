# HBM-side bytes of every kernel of the step: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md 'HBM')
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o fetch -- python3 $R/bench.py --no-graph --serial --no-cpu-baseline --no-alts --steps 20 --warmup 3 > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o write -- python3 $R/bench.py --no-graph --serial --no-cpu-baseline --no-alts --steps 20 --warmup 3 > $R/gpurun_out/pmc_write.log 2>&1
echo done
