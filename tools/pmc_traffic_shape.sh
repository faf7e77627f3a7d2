# HBM-side bytes of every kernel of the step at one shape: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md 'HBM').
# usage: bash tools/pmc_traffic_shape.sh TAG S H L B   ->  gpurun_out/TAG_traffic.json + TAG_traffic.txt
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o fetch -- python3 $R/tools/step_time.py --shape "$@" --steps 4 --reps 1 > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o write -- python3 $R/tools/step_time.py --shape "$@" --steps 4 --reps 1 > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python tools/traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/${TAG}_traffic.json > gpurun_out/${TAG}_traffic.txt
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
cat gpurun_out/${TAG}_traffic.txt
