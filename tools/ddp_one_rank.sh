# One-rank rehearsal of the data-parallel step (real RCCL communicator of one rank; every collective is a copy):
# what the schedule itself costs before a byte crosses xGMI.  Same box, interleaved.  Writes gpurun_out/${TAG}_ddp_one_rank.txt
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    extra = ""
    for k in d:
        if k.startswith("alt_") and isinstance(d[k], dict) and "ms_per_step" in d[k]:
            extra += "  %s %.1f" % (k, d[k]["ms_per_step"] * 1e3)
    print("%-44s %7.1f us/step  %6.2f M frames/s  loss %.6f%s" % (sys.argv[1], d["ms_per_step"] * 1e3, d["value"] / 1e6, d["final_loss"], extra))
except Exception as e:
    print("%-44s FAILED %r" % (sys.argv[1], e))
PY
}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
B="python bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-alts --step-kernels-only"
{
for i in 1 2; do
  $B > $O/tmp_b.json 2>$O/tmp_b.err; line "local step" $O/tmp_b.json
  RV_FORCE_DDP=1 RV_DDP_ALT=0 $B > $O/tmp_b.json 2>$O/tmp_b.err; line "one-rank RCCL, all-reduce, bf16 payload (bench)" $O/tmp_b.json
  RV_FORCE_DDP=1 RV_DDP_ALT=0 RV_DDP_DEFER=0 $B > $O/tmp_b.json 2>$O/tmp_b.err; line "... every step completing itself (RV_DDP_DEFER=0)" $O/tmp_b.json
  RV_FORCE_DDP=1 RV_DDP_ALT=0 RV_DDP_PAYLOAD=fp32 $B > $O/tmp_b.json 2>$O/tmp_b.err; line "one-rank RCCL, all-reduce, fp32 payload (library default)" $O/tmp_b.json
done
} 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_ddp_one_rank.txt
tail -3 $O/tmp_b.err
