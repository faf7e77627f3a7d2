# Round 5, GPU call 3: the paired latent forward (tests, A/B in the step), the deep step's kernel table.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q > $O/r05_gpu_tests_3.txt 2>&1; tail -25 $O/r05_gpu_tests_3.txt
for v in 1 0 1 0 1 0; do RV_LATENT_PAIR=$v python bench.py --no-cpu-baseline --no-alts 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('RV_LATENT_PAIR=$v  %.2f us/step  launches: %s' % (d['ms_per_step'] * 1e3, ' '.join('%.1f' % k['us'] for k in d['kernels'])))"; done > $O/r05_ab_latent_pair.txt; cat $O/r05_ab_latent_pair.txt
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/deep_stats -o s -- python3 $R/tools/deep_bench.py --no-graph --steps 40 --warmup 5 > $O/deep_stats.log 2>&1
cd $R
python - <<'PY' > $O/r05_deep_kernel_stats.txt
import csv, glob
f = glob.glob('gpurun_out/deep_stats/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print('%5.1f%%  calls %4s  avg %7.1f us  %s' % (100 * float(r['TotalDurationNs']) / tot, r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
rm -rf $O/deep_stats; cat $O/r05_deep_kernel_stats.txt
