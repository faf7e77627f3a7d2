run() { python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step']*1e3,1), d['final_loss'])"; }
for rep in 1 2 3; do echo -n "new "; run; echo -n "old "; RV_LIB=$PWD/tools/scratch/lib_old.so run; done
