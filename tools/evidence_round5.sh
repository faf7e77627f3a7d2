# Round-5 evidence, ONE box, one gpurun call: GPU tests, the default bench line, kernel-trace stats / timeline, PMC traffic
# (FETCH_SIZE / WRITE_SIZE passes) and SQ counters of the same command, the data-parallel model (tools/ddp_model.py) and
# one-rank rehearsal, the API-path loop (breakdown with the one-node loss on / off), the real-data loop, the vendor-GEMM
# yardstick next to this build's stand-alone GEMMs, the deep step.  Writes gpurun_out/r05_*; the builder copies them to profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q > $O/r05_gpu_tests.txt 2>&1; tail -3 $O/r05_gpu_tests.txt
python bench.py > $O/r05_bench.json 2> $O/r05_bench.err; tail -c 300 $O/r05_bench.json; echo
bash tools/prof_step.sh r05 > /dev/null 2>&1; head -12 $O/r05_kernel_summary.txt
bash tools/pmc_round.sh r05 > /dev/null 2>&1; tail -12 $O/r05_traffic.txt
bash tools/pmc_sq_step.sh r05 > $O/r05_pmc_sq.log 2>&1; head -14 $O/r05_pmc_sq_summary.txt
{ python tools/ddp_model.py; echo; echo "== the same reference point with stand-in workgroups that take their CUs whole (64 KB of LDS each)"; RV_MODEL_LDS=65536 python tools/ddp_model.py 8 300 15 bf16 | grep "^8\|^(stand"; RV_MODEL_LDS=65536 RV_DDP_W1_WIDE=1 python tools/ddp_model.py 8 300 15 bf16 | grep "^8" | sed 's/$/   <- RV_DDP_W1_WIDE=1/'; RV_DDP_W1_WIDE=1 python tools/ddp_model.py 8 300 15 bf16 | grep "^8" | sed 's/$/   <- RV_DDP_W1_WIDE=1, light stand-in/'; } 2>/dev/null > $O/r05_ddp_model.txt; cat $O/r05_ddp_model.txt
bash tools/ddp_one_rank.sh r05 > /dev/null 2>&1; cat $O/r05_ddp_one_rank.txt
{ for f in 1 0 1 0; do RV_FUSED_LOSS=$f python tools/api_breakdown.py 2>&1 | grep -v amdgpu | sed "s/^/[one-node loss=$f] /"; done; } > $O/r05_api_breakdown.txt; grep "host" $O/r05_api_breakdown.txt
python tools/train_bench.py 2>&1 | grep -v amdgpu | tail -2 > $O/r05_train_bench.txt; cat $O/r05_train_bench.txt
{ python tools/blas_yardstick.py 2>/dev/null; echo; python tools/gemm_bench.py 2>/dev/null; } > $O/r05_blas_yardstick.txt; cat $O/r05_blas_yardstick.txt
for d in fp16 fp32 fp16; do python tools/deep_bench.py --slab-dtype $d 2>/dev/null | tail -1; done > $O/r05_deep_final.txt; cat $O/r05_deep_final.txt
