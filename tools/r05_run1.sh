# Round 5, GPU call 1: the GPU test suite on the round's first batch of changes, the worker's normal-exit variant, the
# default bench line (baseline of this box), SQ counters of the in-step kernels, the deep variant with fp16 / fp32 slabs.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R

python -m pytest tests -m gpu -q -x > $O/r05_gpu_tests_1.txt 2>&1; tail -5 $O/r05_gpu_tests_1.txt
RV_WORKER_HARD_EXIT=0 python -m pytest tests/test_ddp_gpu.py -q -k two_processes > $O/r05_worker_soft_exit.txt 2>&1; tail -3 $O/r05_worker_soft_exit.txt
python bench.py > $O/r05_bench_1.json 2> $O/r05_bench_1.err; tail -c 600 $O/r05_bench_1.json; echo
for d in fp16 fp32 fp16 fp32; do python tools/deep_bench.py --slab-dtype $d 2>/dev/null | tail -1; done > $O/r05_deep_1.txt; cat $O/r05_deep_1.txt
bash tools/pmc_sq_step.sh r05 > $O/r05_pmc_sq.log 2>&1; tail -30 $O/r05_pmc_sq_summary.txt
