for rep in 1 2; do
for args in "" "--serial" "--sched 2" "--no-graph --serial" "--no-graph"; do echo -n "[$args] "; python bench.py $args --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step']*1e3,1), d['config']['launch'])"; done; done
