# Kernel timeline of the one-rank data-parallel step (real RCCL communicator of one rank) under rocprofv3.
# usage: bash tools/prof_ddp.sh TAG [env assignments ...]   -> gpurun_out/TAG_ddp_timeline.txt
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=$1; shift
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RV_FORCE_DDP=1 RV_DDP_ALT=0 RV_DDP_CHECK=0 "$@"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_ddp_prof -o s -- python3 $R/bench.py --no-cpu-baseline --no-alts --step-kernels-only --steps 100 --warmup 10 --repeats 3 > $O/${TAG}_ddp_prof.log 2>&1
cd $R
python tools/timeline_csv.py $(find $O/${TAG}_ddp_prof -name "*kernel_trace.csv" | head -1) 5 > $O/${TAG}_ddp_timeline.txt
rm -rf $O/${TAG}_ddp_prof
cat $O/${TAG}_ddp_timeline.txt
