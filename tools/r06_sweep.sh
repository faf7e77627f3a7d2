#!/bin/bash
# round 6: step time and in-step launch times against the batch (L = 64 and the reference's L = 256), and the four large
# GEMM shapes per tile configuration at B = 131072 -> gpurun_out/r06/{batch_sweep.jsonl,big_batch_gemms.txt}
mkdir -p gpurun_out/r06
python tools/batch_sweep.py > gpurun_out/r06/batch_sweep.jsonl 2> gpurun_out/r06/batch_sweep.err
python - <<'PY'
import json
for l in open("gpurun_out/r06/batch_sweep.jsonl"):
    d = json.loads(l)
    print(d["L"], d["B"], d["us_per_step"], d["step_mfma_frac"], " ".join("%d:%.0f/%.2f" % (k["launch"], k["us"], k["mfma_frac"]) for k in d["kernels"]))
PY
