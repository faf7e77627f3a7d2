# Round 5, GPU call 4: full tests; deep step (slab write-through on / off, fp32); fp8 riders (fc4 only / all) x GEMM tail share;
# the paired latent forward once more (second fc3 slot prefetched across the exchange); the default bench line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q > $O/r05_gpu_tests_4.txt 2>&1; tail -15 $O/r05_gpu_tests_4.txt
{ for k in 1 2; do
    RV_DEEP_SLAB_WT=1 python tools/deep_bench.py --slab-dtype fp16 2>/dev/null | tail -1 | sed 's/^/slab_wt=1 /'
    RV_DEEP_SLAB_WT=0 python tools/deep_bench.py --slab-dtype fp16 2>/dev/null | tail -1 | sed 's/^/slab_wt=0 /'
  done; RV_DEEP_SLAB_WT=1 python tools/deep_bench.py --slab-dtype fp32 2>/dev/null | tail -1 | sed 's/^/slab_wt=1 /'; } > $O/r05_deep_4.txt; cat $O/r05_deep_4.txt
{ for k in 1 2; do for r in fc4 all; do for pct in 0 15 30; do
    RV_FP8_RIDERS=$r RV_WGRAD_TAIL_PCT=$pct python tools/step_time.py --fp8 --tag "fp8 riders=$r gemm-tail-share=$pct%" 2>/dev/null | tail -1
  done; done; python tools/step_time.py --tag "bf16" 2>/dev/null | tail -1; done; } > $O/r05_fp8_riders.txt; cat $O/r05_fp8_riders.txt
{ for v in 1 0 1 0; do RV_LATENT_PAIR=$v python tools/step_time.py --tag "bf16 RV_LATENT_PAIR=$v" 2>/dev/null | tail -1; done; } > $O/r05_ab_latent_pair_2.txt; cat $O/r05_ab_latent_pair_2.txt
python bench.py > $O/r05_bench_4.json 2> $O/r05_bench_4.err; python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r05_bench_4.json') if l.startswith('{')][-1])
print(d['ms_per_step'], [round(k['us'], 1) for k in d['kernels']])
for k in ('alt_fp8', 'alt_deep_c4', 'alt_api_loop', 'alt_fp32_slabs'):
    print(k, {a: b for a, b in d.get(k, {}).items() if 'ms_per' in a or a == 'error'})
PY
