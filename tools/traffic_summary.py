#!/usr/bin/env python3
"""HBM-side bytes per launch of every kernel of the step from the two PMC passes of tools/pmc_round.sh
(rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE runs of `bench.py --step-kernels-only`, CSV).  FETCH_SIZE
(KiB) is doubled as MI355X_MICROARCH.md 'HBM' prescribes for 16-byte-per-lane streaming reads on gfx950; WRITE_SIZE
(KiB) is taken as is.  `per_launch_bytes` is keyed by the launch index bench.py's `kernels` table uses.
    python tools/traffic_summary.py <fetch pass dir> <write pass dir> <out.json>"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
rows = {}
for k in sorted(set(fetch) | set(write)):
    if "at::native" in k or "rocclr" in k:
        continue
    rd, wr = 2.0 * fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
    rows[k[:150]] = {"fetch_size_kib": fetch.get(k, 0.0), "write_size_kib": write.get(k, 0.0),
                     "read_bytes_corrected": rd, "write_bytes": wr, "traffic_bytes": rd + wr, "launches_seen": nf.get(k, 0)}
LAUNCH = (("k_cast_pad_bf16", 0), ("gemm_bf16_kernel<256, 128", 1), ("k_latent_fwd", 2), ("gemm_bf16_kernel<128, 128", 3),
          ("gemm_dgrad_wgrad_kernel", 4), ("k_latent_bwd", 5), ("k_heads_bwd", 6), ("gemm_wgrad_adam_kernel", 7), ("k_adam<true>", 8))
per_launch = {}
for k, v in rows.items():
    for pat, idx in LAUNCH:
        if pat in k:
            per_launch[str(idx)] = v["traffic_bytes"]
dom = [k for k in rows if "gemm_dgrad_wgrad_kernel" in k]
import os
try:
    commit = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".evidence_commit")).read().strip()
except OSError:
    commit = "not recorded"
out = {"method": __doc__.strip(), "commit": commit, "per_kernel": rows, "per_launch_bytes": per_launch,
       "step_total_bytes_one_launch_each": sum(v["traffic_bytes"] for v in rows.values())}
if dom:
    d = rows[dom[0]]
    out.update({"kernel": dom[0], "traffic_bytes": d["traffic_bytes"], "read_bytes_corrected": d["read_bytes_corrected"],
                "write_bytes": d["write_bytes"],
                "algorithmic_bytes": {"read": 29360128, "write": 33554432 + 4 * 32 * 64 * 4,
                                      "note": "dP4, W4, h3 read once (dP4 and h3 feed both GEMMs of the pair), the h3 mask "
                                              "is the same h3; written: dP3 bf16 16.8 MB + dW4 as 4 split-K fp16 slabs "
                                              "16.8 MB + their per-granule exponents (fp32 slabs: 33.5 MB)"}})
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["traffic_bytes"]):
    print("%8.1f MB  (read %7.1f  write %7.1f)  %s" % (v["traffic_bytes"] / 1e6, v["read_bytes_corrected"] / 1e6, v["write_bytes"] / 1e6, k[:110]))
print("sum %.1f MB" % (out["step_total_bytes_one_launch_each"] / 1e6))
