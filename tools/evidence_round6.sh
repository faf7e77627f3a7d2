# Round-6 evidence, ONE box, one gpurun call: GPU tests, the default bench line (with the reference-configuration side lines
# alt_ref_ini / alt_default_ini), kernel-trace stats / timeline of the step at C2 and at the reference's own shape, PMC traffic
# and SQ counters of the default command, the batch sweep with in-step launch tables, the four large GEMMs per tile at
# B = 131072, the latent GEMM forms' K sweep, the data-parallel model and one-rank rehearsal, the deep step.
# Writes gpurun_out/r06_*; the builder copies them to profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
# (.evidence_commit is written by the caller before the snapshot travels: the box has no .git)
python -m pytest tests -m gpu -q -s > $O/r06_gpu_tests_full.txt 2>&1; grep -v "amdgpu.ids" $O/r06_gpu_tests_full.txt | grep "DDP_BF16\|passed\|failed" > $O/r06_gpu_tests.txt; cat $O/r06_gpu_tests.txt
python bench.py > $O/r06_bench.json 2> $O/r06_bench.err; tail -c 400 $O/r06_bench.json; echo
bash tools/prof_step.sh r06 > /dev/null 2>&1; head -14 $O/r06_kernel_summary.txt
bash tools/r06_prof_shape.sh r06_refini 1024 2048 256 4096 > $O/r06_refini_timeline.txt 2>&1; head -12 $O/r06_refini_timeline.txt
bash tools/r06_prof_shape.sh r06_default_ini 1024 2048 256 131072 > $O/r06_default_ini_timeline.txt 2>&1; head -16 $O/r06_default_ini_timeline.txt
bash tools/pmc_round.sh r06 > /dev/null 2>&1; tail -12 $O/r06_traffic.txt
bash tools/pmc_sq_step.sh r06 > $O/r06_pmc_sq.log 2>&1; head -14 $O/r06_pmc_sq_summary.txt
bash tools/pmc_sq_shape.sh r06_refini 1024 2048 256 4096 > /dev/null 2>&1; head -16 $O/r06_refini_pmc_sq.txt
bash tools/pmc_sq_shape.sh r06_default_ini 1024 2048 256 131072 > /dev/null 2>&1; head -16 $O/r06_default_ini_pmc_sq.txt
bash tools/pmc_traffic_shape.sh r06_default_ini 1024 2048 256 131072 > /dev/null 2>&1; cat $O/r06_default_ini_traffic.txt
bash tools/r06_sweep.sh > $O/r06_batch_sweep_table.txt 2>&1; cp $O/r06/batch_sweep.jsonl $O/r06_batch_sweep.jsonl; cat $O/r06_batch_sweep_table.txt
python tools/big_batch_gemms.py 131072 2>&1 | grep -v amdgpu > $O/r06_big_batch_gemms.txt; cat $O/r06_big_batch_gemms.txt
python tools/latent_k_sweep.py 2>&1 | grep -v amdgpu > $O/r06_latent_k_sweep.txt; cat $O/r06_latent_k_sweep.txt
python tools/large_batch_k_sweep.py 2>&1 | grep -v amdgpu > $O/r06_large_batch_k_sweep.txt; cat $O/r06_large_batch_k_sweep.txt
{ python tools/ddp_model.py; echo; echo "== the same reference point with stand-in workgroups that take their CUs whole (64 KB of LDS each)"; RV_MODEL_LDS=65536 python tools/ddp_model.py 8 300 15 bf16 | grep "^8\|^(stand"; } 2>/dev/null > $O/r06_ddp_model.txt; cat $O/r06_ddp_model.txt
bash tools/ddp_one_rank.sh r06 > /dev/null 2>&1; cat $O/r06_ddp_one_rank.txt
for i in 1 2 3; do python tools/deep_bench.py 2>/dev/null | tail -1; done > $O/r06_deep.txt; cat $O/r06_deep.txt
