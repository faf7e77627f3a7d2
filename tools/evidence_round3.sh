# Round-3 evidence, ONE box, one gpurun call: the default bench line, the kernel-trace stats / timeline of the same
# command, every same-box A/B DESIGN.md section 6 quotes, the one-rank data-parallel rehearsal and the real-data loop.
# Writes gpurun_out/r03_*; the builder copies them to profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-40s %7.1f us/step  %6.2f M frames/s  pair %5.1f us (frac %.3f)  loss %.6f" % (sys.argv[1], d["ms_per_step"] * 1e3, d["value"] / 1e6, d["roofline"]["us_per_launch"], d["roofline"]["frac"], d["final_loss"]))
PY
}
B="python bench.py --steps 200 --warmup 20 --repeats 7 --no-cpu-baseline --no-alts"
{
echo "== same box, interleaved twice; default = fp16 block-floating-point slabs, row-local latent forward and backward, eager launches"
for i in 1 2; do
  $B > $O/tmp_b.json 2>/dev/null; line "default" $O/tmp_b.json
  $B --slab-dtype fp32 > $O/tmp_b.json 2>/dev/null; line "fp32 split-K slabs (round 2 default)" $O/tmp_b.json
  $B --latent-fused 0 > $O/tmp_b.json 2>/dev/null; line "latent fwd / bwd as 3 + 3 launches" $O/tmp_b.json
  $B --slab-dtype fp32 --latent-fused 0 > $O/tmp_b.json 2>/dev/null; line "both (round 2's step + epilogue prefetch)" $O/tmp_b.json
  $B --graph-pool > $O/tmp_b.json 2>/dev/null; line "one hipGraph of 8 steps (--graph-pool)" $O/tmp_b.json
  $B --graph > $O/tmp_b.json 2>/dev/null; line "one hipGraph per step (--graph)" $O/tmp_b.json
done
} 2>&1 | grep -v amdgpu.ids > $O/r03_ab_step.txt
cat $O/r03_ab_step.txt
bash tools/ddp_one_rank.sh > /dev/null 2>&1 || true
cat $O/r03_ddp_one_rank.txt
python tools/train_bench.py 2>&1 | grep -v amdgpu | tail -2 > $O/r03_train_bench.txt; cat $O/r03_train_bench.txt
bash tools/prof_round3.sh r03_final > /dev/null
cat $O/r03_final_timeline.txt
python bench.py > $O/r03_bench.json 2> $O/r03_bench.err
tail -c 600 $O/r03_bench.json
