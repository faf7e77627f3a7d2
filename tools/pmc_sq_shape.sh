# SQ counters of every kernel of the training step at ANY shape (the 8 SQ slots in one rocprofv3 --pmc pass over
# tools/step_time.py, python3 directly behind `--`): MFMA busy, waves parked / stalled / issuing, LDS conflicts per kernel.
# usage: bash tools/pmc_sq_shape.sh TAG S H L B   ->  gpurun_out/TAG_pmc_sq.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=$1; shift
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
cd /tmp
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_$TAG -o sq -- python3 $R/tools/step_time.py --shape "$@" --steps 40 --reps 1 > $O/pmc_sq_$TAG.log 2>&1
cd $R
python3 - $O/pmc_sq_$TAG "$@" > $O/${TAG}_pmc_sq.txt <<'PY'
import collections, csv, glob, sys
d = sys.argv[1]
S, H, L, B = (int(v) for v in sys.argv[2:6])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# SQ counters per kernel of the training step, S=%d H=%d L=%d B=%d (one rocprofv3 --pmc pass: dispatches are serialised and the" % (S, H, L, B))
print("# clock differs from the timed runs; `us` is the duration IN THIS PASS).  mfma us = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / 2.4 GHz;")
print("# busy = mfma us / us; wait / stall / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES.")
print("%-8s %9s %9s %6s | %5s %5s %6s | %8s  %s" % ("calls", "us", "mfma us", "busy", "wait", "stall", "active", "lds_conf", "kernel"))
rows = []
for k, c in acc.items():
    if "vectorized" in k or "rocclr" in k or "index_elementwise" in k or "distribution" in k:
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    us = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else 0.0
    mf = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024 / 2.4e3
    wc = max(m.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    rows.append((us, len(dur[k]), mf, m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                 m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0), 1.0), k))
for us, n, mf, w, s, a, lc, k in sorted(rows, reverse=True):
    print("%-8d %9.1f %9.1f %5.1f%% | %4.0f%% %4.0f%% %5.0f%% | %7.2f%%  %s" % (n, us, mf, 100 * mf / max(us, 1e-9), 100 * w, 100 * s, 100 * a, 100 * lc, k[:110]))
PY
rm -rf $O/pmc_sq_$TAG
cat $O/${TAG}_pmc_sq.txt
