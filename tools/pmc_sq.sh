# SQ-side counters (issue/wait split, LDS conflicts, MFMA busy) per GEMM of the step, stand-alone (tools/gemm_bench.py)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -o sq -- python3 $R/tools/gemm_bench.py > $R/gpurun_out/pmc_sq.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_sq gemm > gpurun_out/pmc_sq_summary.txt 2>&1
echo done
