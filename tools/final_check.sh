R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q > $O/r05_gpu_tests_final.txt 2>&1; tail -4 $O/r05_gpu_tests_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --no-cpu-baseline > $O/r05_bench_final.json 2> $O/r05_bench_final.err; python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r05_bench_final.json') if l.startswith('{')][-1])
print(d['ms_per_step'], [round(k['us'], 1) for k in d['kernels']])
for k in ('alt_fp8', 'alt_deep_c4', 'alt_api_loop'):
    print(k, {a: round(b, 4) for a, b in d.get(k, {}).items() if 'ms_per' in a})
PY
bash tools/pmc_sq_step.sh r05 > /dev/null 2>&1; head -12 $O/r05_pmc_sq_summary.txt
