# L2 (TCC) request counters of every launch of the step: how many of a launch's read requests from the CUs (TCP_TCC_READ_REQ,
# 128 B each) the L2s had to fetch from the fabric (TCC_EA0_RDREQ) -- the first-touch share DESIGN.md section 6 derives the K
# loops' rate from.  Two rocprofv3 --pmc passes over `bench.py --step-kernels-only`, python3 directly behind `--`.
# usage: bash tools/pmc_tcc.sh  ->  gpurun_out/r05_tcc_hits.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
for PC in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $PC | cut -d' ' -f1)
  rocprofv3 --pmc $PC --kernel-trace --output-format csv -d $O/pmc_tcc_$tag -o t -- python3 $R/bench.py --no-cpu-baseline --no-alts --step-kernels-only --steps 10 --warmup 2 --repeats 1 > $O/pmc_tcc_$tag.log 2>&1 || echo "pass $tag failed"
done
cd $R
python - <<'PY' > gpurun_out/r05_tcc_hits.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_tcc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    if "vectorized" in k or "copyBuffer" in k: continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
    print("%-72s hit %12.0f miss %12.0f  hit rate %.3f  req %12.0f read %12.0f | EA rd %10.0f (32B %8.0f) wr %10.0f  TCP->TCC rd %12.0f" % (
        k, hit, miss, hit / max(hit + miss, 1), m.get("TCC_REQ_sum", 0), m.get("TCC_READ_sum", 0), m.get("TCC_EA0_RDREQ_sum", 0),
        m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_EA0_WRREQ_sum", 0), m.get("TCP_TCC_READ_REQ_sum", 0)))
PY
cat gpurun_out/r05_tcc_hits.txt; tail -3 $O/pmc_tcc_TCC_HIT_sum.log
rm -rf $O/pmc_tcc_*/
