#!/usr/bin/env python3
"""Per-kernel SQ counters of the training step's launches, from two rocprofv3 --pmc passes over the same command (the
8 SQ slots; GRBM_GUI_ACTIVE for the clock) -- tools/pmc_sq_step.sh.

    python tools/pmc_sq_summary.py <SQ pass dir> <GRBM pass dir> bf16|fp8

Columns per kernel (averages over its dispatches in the pass):
  us         dispatch duration IN THE PMC PASS (counters serialise dispatches and the chip clocks differently: the bench's
             own in-step figure is in BENCH / profiles/*_bench.json)
  mfma us    SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / 2.4 GHz: how long the matrix pipe of an average SIMD was busy, at the
             clock the dense peak assumes
  busy       mfma us / us: the share of the dispatch the matrix pipes were busy (at 2.4 GHz; the chip holds less under
             load, so the true share of CYCLES is higher by clock_nominal / clock_held)
  busy@grbm  the same against the dispatch's own cycles, GRBM_GUI_ACTIVE / 8 XCDs (reads HIGH on dispatches this short --
             MI355X_MICROARCH.md, DVFS give-back -- so this column is a LOWER bound of the share of cycles)
  flops/t    the launch's algorithmic FLOPs / duration / dense peak (2.5 PF bf16; 5 PF where the launch's large GEMM runs on
             fp8 operands) -- the figure the bench prints.  busy == flops/t means every MFMA issued is algorithmic work
             (no recomputation) and the launch's distance from the peak is matrix-pipe IDLE time, which the next columns split
  of the wave-cycles (SQ_WAVE_CYCLES): wait = SQ_WAIT_ANY (parked at s_waitcnt / barrier), stall = SQ_WAIT_INST_ANY (issue
  stalls, of which lds = SQ_WAIT_INST_LDS), active = SQ_ACTIVE_INST_ANY
  lds_conf   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import csv
import glob
import sys

B, S, H, L = 4096, 1024, 2048, 64
GF = 1e9
# (pattern, label, algorithmic GFLOP per launch, runs on fp8 operands in the fp8 path)
KERNELS = (
    ("k_cast_pad_bf16", "0 cast", 0.0, False), ("k_cast_pad_bf16_q8", "0 cast + quantise", 0.0, False),
    ("gemm_bf16_kernel<256, 128", "1 fc1 forward 256x128", 2.0 * B * S * H / GF, True),
    ("k_latent_fwd", "2 latent forward (heads + reparam + fc3)", (2.0 * B * H * 2 * L + 2.0 * B * L * H) / GF, False),
    ("gemm_bf16_kernel<128, 128", "3 fc4 forward + loss 128x128", 2.0 * B * H * S / GF, True),
    ("gemm_dgrad_wgrad_kernel", "4 paired fc4 backward 256x256", 4.0 * B * H * S / GF, True),
    ("k_latent_bwd", "5 latent backward (dz + reparam' + dW3)", 4.0 * B * H * L / GF, False),
    ("k_heads_bwd", "6 heads backward (dP1 + dWh)", 8.0 * B * H * L / GF, False),
    ("gemm_wgrad_adam_kernel", "7 dW1 + optimizer riders", 2.0 * B * S * H / GF, True),
    ("k_adam<true>", "8 Adam(fc1)", 0.0, False),
    ("k_fp8_wmax", "9 fp8 weight maxima", 0.0, False),
)


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (k, r.get("Dispatch_Id"))
            if key not in seen and "End_Timestamp" in r and r.get("End_Timestamp"):
                seen.add(key)
                dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    return acc, dur


def main():
    sq, dur = load(sys.argv[1])
    clk, dclk = load(sys.argv[2])
    fp8 = len(sys.argv) > 3 and sys.argv[3] == "fp8"
    av = lambda v: sum(v) / len(v) if v else 0.0  # noqa: E731
    print("%-46s %5s %7s %7s %6s %9s %8s | %5s %5s %5s %6s | %8s" % ("launch", "n", "us", "mfma us", "busy", "busy@grbm", "flops/t", "wait",
                                                                        "stall", "(lds)", "active", "lds_conf"))
    rows = []
    for name in sq:
        hit = [k for k in KERNELS if k[0] in name]
        if not hit:
            continue
        pat, label, gflop, is8 = max(hit, key=lambda k: len(k[0]))
        c = {n: av(v) for n, v in sq[name].items()}
        us = av(dur[name])
        cyc = av(clk.get(name, {}).get("GRBM_GUI_ACTIVE", [])) / 8.0       # cycles of the dispatch (sum over 8 XCDs / 8)
        us_c = av(dclk.get(name, []))
        ghz = cyc / (us_c * 1e3) if us_c else 0.0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc else float("nan")
        peak = 5.0e15 if (fp8 and is8) else 2.5e15
        ft = gflop * 1e9 / (us * 1e-6) / peak if us else 0.0
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        lds_a = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        mfma_us = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / 2.4e3
        if fp8 and is8:
            mfma_us *= 1.0      # (an MX-scaled fp8 MFMA is busy twice the cycles of the bf16 form at 4x the K: the counter has it)
        rows.append((label, len(dur[name]), us, (mfma_us, mfma_us / us if us else 0.0, busy), busy, ft, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
                     c.get("SQ_WAIT_INST_LDS", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                     c.get("SQ_LDS_BANK_CONFLICT", 0) / lds_a if lds_a else 0.0, name))
    for r in sorted(rows):
        print("%-46s %5d %7.1f %7.1f %5.1f%% %8.1f%% %7.1f%% | %4.0f%% %4.0f%% %4.0f%% %5.0f%% | %7.2f%%   %s" % (
            r[0], r[1], r[2], r[3][0], 100 * r[3][1], 100 * r[3][2], 100 * r[5], 100 * r[6], 100 * r[7], 100 * r[8], 100 * r[9], 100 * r[10],
            r[11][:70]))


if __name__ == "__main__":
    main()
