# Round-2 A/B evidence, one box, one call: every comparison DESIGN.md section 6 quotes.  Writes gpurun_out/r02_*.txt
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-28s %7.1f us/step  %6.2f M frames/s  pair %5.1f us (frac %.3f)  loss %.6f" % (sys.argv[1], d["ms_per_step"] * 1e3, d["value"] / 1e6, d["roofline"]["us_per_launch"], d["roofline"]["frac"], d["final_loss"]))
PY
}
{
echo "== schedules (bench.py --sched: 0 default = fc1 wgrad launch carries Adam(fc3,fc4); 3 = round-1 single stream; 2 = two streams), interleaved twice"
for i in 1 2; do for s in 3 0 2; do python bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-alts --sched $s > $O/tmp_b.json 2>/dev/null; line "sched $s" $O/tmp_b.json; done; done
echo "== split-K slab dtype of dW1/dW4"
for i in 1 2; do for d in fp32 fp16; do python bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-alts --slab-dtype $d > $O/tmp_b.json 2>/dev/null; line "slabs $d" $O/tmp_b.json; done; done
echo "== 256x128 main loop (3 ring = default, 9 ping-pong)"
for i in 1 2; do for l in 3 9; do python bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-alts --n128-loop $l > $O/tmp_b.json 2>/dev/null; line "n128 loop $l" $O/tmp_b.json; done; done
} > $O/r02_ab_step.txt 2>&1
{
for l in 3 9 3 9; do echo "== 256x128 main loop $l (tools/gemm_bench.py, stand-alone GEMMs, split-K 4 weight gradients)"; RV_N128=$l SPL_w1=4 SPL_w4=4 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids; done
} > $O/r02_ab_gemm.txt 2>&1
python tools/gemm_decomp.py 2>&1 | grep -v amdgpu.ids > $O/r02_gemm_decomp.txt
python tools/train_bench.py 2>&1 | tail -2 > $O/r02_train_bench.txt
python tools/deep_bench.py 2>&1 | tail -3 > $O/r02_deep_bench.txt || true
python tools/api_bench.py 2>&1 | tail -1 > $O/r02_api_bench.txt || true
cat $O/r02_ab_step.txt $O/r02_train_bench.txt $O/r02_deep_bench.txt $O/r02_api_bench.txt
