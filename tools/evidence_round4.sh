# Round-4 evidence, ONE box, one gpurun call: GPU tests, the default bench line, kernel-trace stats / timeline and PMC
# traffic of the same command, the data-parallel model (tools/ddp_model.py) and one-rank rehearsal with its timeline,
# the API-path loop, the real-data loop.  Writes gpurun_out/r04_*; the builder copies them to profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q > $O/r04_gpu_tests.txt 2>&1; tail -2 $O/r04_gpu_tests.txt
python bench.py > $O/r04_bench.json 2> $O/r04_bench.err; tail -c 400 $O/r04_bench.json; echo
bash tools/prof_step.sh r04 > /dev/null 2>&1; cat $O/r04_kernel_summary.txt | head -10
bash tools/pmc_round.sh r04 > /dev/null 2>&1; tail -11 $O/r04_traffic.txt
{ python tools/ddp_model.py; echo; echo "== the same reference point with stand-in workgroups that take their CUs whole (64 KB of LDS each)"; RV_MODEL_LDS=65536 python tools/ddp_model.py 8 300 15 bf16 | grep "^8\|^(stand"; RV_MODEL_LDS=65536 RV_DDP_W1_WIDE=1 python tools/ddp_model.py 8 300 15 bf16 | grep "^8" | sed 's/$/   <- RV_DDP_W1_WIDE=1/'; RV_DDP_W1_WIDE=1 python tools/ddp_model.py 8 300 15 bf16 | grep "^8" | sed 's/$/   <- RV_DDP_W1_WIDE=1, light stand-in/'; echo "== cross-stream edges as HIP events instead of device-side flags"; RV_DDP_SIGNAL=event python tools/ddp_model.py 8 300 15 bf16 | grep "^8"; } 2>/dev/null > $O/r04_ddp_model.txt; cat $O/r04_ddp_model.txt
bash tools/ddp_one_rank.sh r04 > /dev/null 2>&1; cat $O/r04_ddp_one_rank.txt
bash tools/prof_ddp.sh r04 > /dev/null 2>&1; cat $O/r04_ddp_timeline.txt
{ for h in 1 0 1 0; do RV_OPTIM_HOOK=$h python tools/api_prof.py 2>&1 | grep "100 steps" | sed "s/$/  RV_OPTIM_HOOK=$h/"; done; python tools/api_bench.py 2>&1 | tail -1; } > $O/r04_api_bench.txt; cat $O/r04_api_bench.txt
python tools/train_bench.py 2>&1 | grep -v amdgpu | tail -2 > $O/r04_train_bench.txt; cat $O/r04_train_bench.txt
