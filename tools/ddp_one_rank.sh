# One-rank rehearsal of the data-parallel step (real RCCL communicator of one rank; every collective is a copy):
# what the schedule itself costs before a byte crosses xGMI.  Same box, interleaved.  Writes gpurun_out/r03_ddp_one_rank.txt
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    extra = ""
    for k in ("alt_sharded", "alt_allreduce", "alt_allreduce_bf16_payload", "alt_bf16_payload"):
        if k in d and "ms_per_step" in d[k]:
            extra += "  %s %.1f" % (k, d[k]["ms_per_step"] * 1e3)
    print("%-34s %7.1f us/step  %6.2f M frames/s  loss %.6f%s" % (sys.argv[1], d["ms_per_step"] * 1e3, d["value"] / 1e6, d["final_loss"], extra))
except Exception as e:
    print("%-34s FAILED %r" % (sys.argv[1], e))
PY
}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
{
for i in 1 2; do
  python bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-alts > $O/tmp_b.json 2>$O/tmp_b.err; line "local step" $O/tmp_b.json
  RV_FORCE_DDP=1 RV_DDP_MODE=allreduce RV_DDP_ALT=0 python bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-alts > $O/tmp_b.json 2>$O/tmp_b.err; line "one-rank RCCL, all-reduce schedule" $O/tmp_b.json
  RV_FORCE_DDP=1 RV_DDP_MODE=sharded RV_DDP_ALT=0 python bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-alts > $O/tmp_b.json 2>$O/tmp_b.err; line "one-rank RCCL, sharded optimizer" $O/tmp_b.json
done
} 2>&1 | grep -v amdgpu.ids | tee $O/r03_ddp_one_rank.txt
tail -3 $O/tmp_b.err
