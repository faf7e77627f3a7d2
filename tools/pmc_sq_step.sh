# SQ-side counters of the kernels AS THEY RUN IN THE STEP (round-4 verdict item 5): one pass of the 8 SQ slots and one of
# GRBM_GUI_ACTIVE (the clock) over `bench.py --step-kernels-only` (bf16 headline step) and over tools/fp8_steps.py (the fp8
# weight path), python3 directly behind `--` (the profiler's preloaded library has initialised the GPU).  Collected in
# runs of their own: PMC passes serialise kernels and lower the clock, their durations are not the bench's.
# usage: bash tools/pmc_sq_step.sh TAG  ->  gpurun_out/TAG_pmc_sq_summary.txt
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${1:-r05}
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
cd /tmp
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_bf16 -o sq -- python3 $R/bench.py --no-cpu-baseline --no-alts --step-kernels-only --steps 20 --warmup 3 --repeats 1 > $O/pmc_sq_bf16.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_clk_bf16 -o clk -- python3 $R/bench.py --no-cpu-baseline --no-alts --step-kernels-only --steps 20 --warmup 3 --repeats 1 > $O/pmc_clk_bf16.log 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_fp8 -o sq -- python3 $R/tools/fp8_steps.py 60 > $O/pmc_sq_fp8.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_clk_fp8 -o clk -- python3 $R/tools/fp8_steps.py 60 > $O/pmc_clk_fp8.log 2>&1
cd $R
{ echo "# bf16 step (bench.py --step-kernels-only), commit $(cat $R/.evidence_commit 2>/dev/null || echo unknown)"; python tools/pmc_sq_summary.py $O/pmc_sq_bf16 $O/pmc_clk_bf16 bf16;
  echo; echo "# fp8 weight path (tools/fp8_steps.py)"; python tools/pmc_sq_summary.py $O/pmc_sq_fp8 $O/pmc_clk_fp8 fp8; } > $O/${TAG}_pmc_sq_summary.txt
rm -rf $O/pmc_sq_bf16 $O/pmc_clk_bf16 $O/pmc_sq_fp8 $O/pmc_clk_fp8
cat $O/${TAG}_pmc_sq_summary.txt
