timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/full.log 2>&1; tail -3 gpurun_out/full.log
for rep in 1 2 3; do python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step']*1e3,1), d['config']['launch'])"; done
python tools/deep_bench.py 2>&1 | tail -1 | cut -c1-200
