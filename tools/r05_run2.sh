# Round 5, GPU call 2: the one-node loss path and the deep engine's riders.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests/test_model_gpu.py tests/test_deep_gpu.py tests/test_train_entry.py -m gpu -q -x > $O/r05_gpu_tests_2.txt 2>&1; tail -15 $O/r05_gpu_tests_2.txt
for d in fp16 fp32 fp16; do python tools/deep_bench.py --slab-dtype $d 2>/dev/null | tail -1; done > $O/r05_deep_2.txt; cat $O/r05_deep_2.txt
{ for f in 1 0 1 0; do RV_FUSED_LOSS=$f python tools/api_breakdown.py 2>&1 | grep -v amdgpu | sed "s/^/[RV_FUSED_LOSS=$f] /"; done; python tools/api_bench.py 2>&1 | tail -1; } > $O/r05_api_2.txt; cat $O/r05_api_2.txt
