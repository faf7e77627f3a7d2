#!/usr/bin/env python3
"""Benchmark of the MI355X training path: audio frames/sec for a full training step (forward + fused loss + backward +
Adam) on synthetic 1024-sample frames.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] / configs[2]): S=1024, H=2048, L=64, per-GPU batch 4096, bf16 MFMA inputs with fp32
accumulation, fp32 master weights and Adam state.  A step is one pass of the hot path over one resident batch (a pool of
8 distinct device-resident batches is cycled); eps is drawn on-device.  One host call per step (`rv_plan_step`) enqueues
its 9 kernels back to back on one stream, eagerly (`--graph` replays them from a hipGraph instead: the kernels run as
fast, consecutive replays are a few us apart).

N > 1: one process per GPU; each step is one `rv_plan_step_ddp` call that also issues the RCCL collectives (all-reduce
schedule, two buckets; the bf16 gradient payload by explicit choice of this bench -- named in `config.ddp_payload`, with
the library's default, the exact fp32 mean, timed beside it as `alt_fp32_payload`: DESIGN.md section 5;
`RV_DDP_PAYLOAD=fp32` selects the exact mean; `alt_fp32_payload` and `alt_no_defer` are timed beside the headline).  Weak scaling: per-GPU batch fixed.  Before it is
timed the library-driven step is checked on scratch engines against the torch.distributed route; after every section
that can fail on one rank alone the ranks AGREE on success (a MIN all-reduce) -- if any rank failed inside a step, every
rank exits non-zero instead of going on to a collective its peers will never join.

Rank 0 prints ONE JSON line.  `roofline` is for the LONGEST launch of the step, found in this run: every launch of the
step is timed IN the step with HIP events (a hipGraph of steps minus the same graph without that launch; `kernels` holds
all nine rows).  `cpu_baseline` is the stock-PyTorch CPU port of the same step (oracle/torch_port.py) timed on this node's
host cores (N=1 only).  Side lines at N=1 (never the headline; `--no-alts` skips them): `alt_fp8` / `alt_fp8_forward_only`
(the fp8 weight path), `alt_ref_ini` / `alt_default_ini` (the reference's own model, latent_dim 256, at batch 4096 and at
default.ini's 131072, with every launch timed in the step), `alt_fp32_slabs`, `alt_deep_c4` (BASELINE configs[3]) and `alt_api_loop` (the reference's loop
unchanged through `rawvae.model` + `torch.optim.Adam`; host-bound).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

S, H, L, B = 1024, 2048, 64, 4096
KL_BETA, LR = 1e-4, 1e-4
POOL = 8
PEAK_BF16_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_HBM_GBS = 8000.0      # same guide, "HBM3E peak BW" (6.29 TB/s measured for a float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graph", action="store_true",
                    help="replay each step from a hipGraph (N=1); a few us/step slower than back-to-back eager launches "
                         "of the same kernels (gap between replays), so eager is the default")
    ap.add_argument("--graph-pool", action="store_true",
                    help="(N=1) ONE hipGraph holding the steps of the whole batch pool (8 steps per replay), so the gap "
                         "between replays is paid once per 8 steps; steps beyond a multiple of 8 run eagerly")
    ap.add_argument("--no-graph", action="store_true", help="(default) eager launches")
    ap.add_argument("--latent-fused", type=int, default=None, choices=[0, 1],
                    help="heads + reparam + fc3 of the forward as one row-local launch (1, default) or three launches (0)")
    ap.add_argument("--fp8", action="store_true",
                    help="ignored (kept for old command lines): the headline is bf16; the fp8 path is timed as the side line `alt_fp8`")
    ap.add_argument("--step-kernels-only", action="store_true",
                    help="profiling runs (tools/prof_round.sh, tools/pmc_round.sh): launch nothing but full training steps -- "
                         "no per-kernel differential timing (its graphs leave launches out), no side lines; `roofline` is "
                         "then null and the profile's own per-kernel durations are the figures")
    ap.add_argument("--no-alts", action="store_true",
                    help="skip the side lines `alt_fp8`, `alt_deep_c4`, `alt_fp32_slabs` and `alt_api_loop`; they are timed after the "
                         "headline at N=1 and never replace it")
    ap.add_argument("--slab-dtype", default=None, choices=["fp32", "fp16"],
                    help="element type of the fc1 / fc4 weight-gradient split-K slabs (default: the engine's)")
    ap.add_argument("--roctx", action="store_true", help="roctx ranges around the step's phases (rocprofv3 --marker-trace)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed passes of K steps each (0 = at least 5 and enough for --min-seconds of timed work)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="minimum total timed work (all repeats)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------- per-launch accounting
def step_launches(eng):
    """The launches of one training step in issue order, grouped as the plan groups them (rv_plan_diag_skip bit k leaves
    group k out): (name, algorithmic FLOPs, algorithmic HBM bytes).  Any model / batch shape: at a padded latent width
    of 64 (C2) group 2 is the row-local k_latent_fwd, above it (the reference's own latent_dim = 256) the heads GEMM with
    the reparameterisation in its epilogue followed by fc3's GEMM; groups 5 / 6 likewise.  FLOPs are SURVEY 8d's 2 x MACs
    of the contractions in the group; bytes are every operand read once and every output written once at the element
    types the step uses (bf16 activations, fp32 frames / latent tensors, fp16 block-floating-point slabs for dW1 / dW4,
    fp32 slabs elsewhere; Adam: 12 B read + 12 B written + 2 B shadow per parameter plus its gradient slabs)."""
    S, H, L, B = eng.S, eng.H, eng.L, eng.B
    Bp, Sp, Hp, Lp = eng.padded()
    sb = 2 if eng.slab_dtype == "fp16" else 4
    descs = eng.plan_descs()

    def adam_bytes(ds):
        n = 0
        for d in ds:
            el = d.rows * d.cols
            n += el * (26 if d.shadow_bf16 else 28) + el * d.grad_splits * (2 if d.grad_half else 4)
        return n
    s_w1, s_w4 = descs[0].grad_splits, descs[8].grad_splits
    rf, rl = eng.riders()   # tensors [rf, rl) are updated beside fc1's weight gradient, the others by the last launch
    names = ["fc1", "fc1", "fc21", "fc21", "fc22", "fc22", "fc3", "fc3", "fc4", "fc4"]
    riders = ", ".join(dict.fromkeys(names[rf:rl]))
    tail = ", ".join(dict.fromkeys(names[:rf] + names[rl:]))
    d_riders, d_tail = list(descs[rf:rl]), list(descs[:rf]) + list(descs[rl:])
    dims = "%dx%dx%d" % (B, H, S)
    riders_on = Hp % 256 == 0 and Sp % 256 == 0 and (Hp // 256) * (Sp // 256) * s_w1 <= 192   # plan.hip's full-local schedule
    rowlocal = Lp == 64 and Hp % 512 == 0 and Hp <= 2048 and Bp <= 8192   # csrc/latent.hip rv_latent_rowlocal
    rows = [
        ("k_cast_pad_bf16 (frames fp32 -> padded bf16 operand)", 0.0, B * S * 4 + Bp * Sp * 2),
        ("fc1 forward GEMM: relu(x W1^T + b1) %s" % dims, 2.0 * B * S * H,
         Bp * Sp * 2 + Hp * Sp * 2 + Bp * Hp * 2),
        (("k_latent_fwd: heads GEMM + reparameterisation + KL partials + fc3 (row-local)" if rowlocal else
          "k_heads_reparam_gemm (heads GEMM, 64x128 tiles -- 256x128 at large batches --, reparameterisation + KL partials in the "
          "epilogue) + fc3 forward GEMM"),
         2.0 * B * H * 2 * L + 2.0 * B * L * H,
         Bp * Hp * 2 + Bp * Hp * 2 + (2 * Lp * Hp + Hp * Lp) * 2 + Bp * 2 * Lp * 4 + Bp * Lp * (4 + 2) + (0 if rowlocal else Bp * Lp * 2)),
        ("fc4 forward GEMM + tanh + MSE partials + dP4: %dx%dx%d" % (B, S, H), 2.0 * B * H * S,
         Bp * Hp * 2 + Sp * Hp * 2 + B * S * 4 + Bp * Sp * 2),
        ("fc4 backward: dX=relu'(dY W) %s + dW=dY^T X %dx%dx%d split-K %d, %s slabs (one paired 256x256 launch where the extents "
         "allow)" % (dims, S, H, B, s_w4, eng.slab_dtype), 4.0 * B * H * S,
         Bp * Sp * 2 + Sp * Hp * 2 + Bp * Hp * 2 + Bp * Hp * 2 + s_w4 * Sp * Hp * sb),
        (("k_latent_bwd: dz = dP3 W3 + reparameterisation backward + dW3 (co-resident workgroups)" if rowlocal else
          "k_dz_reparam_gemm: dz = dP3 W3 on 64x64 tiles (256x128 at large batches) with the reparameterisation backward in the "
          "epilogue + dW3 on 128x128 tiles, one launch"), 4.0 * B * H * L,
         Bp * Hp * 2 + Hp * Lp * 2 + Bp * Lp * 2 + Bp * 2 * Lp * (4 + 2) + Bp * Lp * 4 + descs[6].grad_splits * Hp * Lp * 4),
        (("k_heads_bwd: dP1 = relu'(dmulv Wh) + dWh, one pass over h1" if rowlocal else
          "heads backward: dP1 = relu'(dmulv Wh) + dWh = dmulv^T h1"), 4.0 * B * H * 2 * L,
         Bp * Hp * 2 + Bp * 2 * Lp * 2 + Bp * Hp * 2 + descs[2].grad_splits * 2 * Lp * Hp * (2 if descs[2].grad_half else 4)),
        # (the rider blocks exist where fc1's weight gradient leaves CUs idle: 256 x 256 tiles x K splits <= 192 blocks; at large
        # batches it fills the chip and the whole optimizer is the last launch)
        (("fc1 weight gradient dW=dY^T X %dx%dx%d split-K %d (%s slabs) + Adam of %s beside it (rider blocks)" % (
            H, S, B, s_w1, eng.slab_dtype, riders)) if riders_on else
         ("fc1 weight gradient dW=dY^T X %dx%dx%d split-K %d (%s slabs), 256x256 tiles on all CUs" % (H, S, B, s_w1, eng.slab_dtype)),
         2.0 * B * S * H, Bp * Hp * 2 + Bp * Sp * 2 + s_w1 * Hp * Sp * sb + (adam_bytes(d_riders) if riders_on else 0)),
        ("k_adam<true> Adam of %s (sums the gradient slabs, %d of dW1; refreshes the bf16 shadows)" % (
            tail if riders_on else "all ten tensors", s_w1), 0.0, adam_bytes(d_tail if riders_on else descs[0:10])),
    ]
    return rows


def time_launches_in_step(eng, x, steps=10, reps=7):
    """Duration of EVERY launch of the step, IN the step, with HIP events on the launching stream: for launch k, two
    hipGraphs of `steps` full training steps -- the plan's own schedule, one of them with launch k left out
    (rv_plan_diag_skip, include/rawvae_hip_diag.h) -- are replayed alternately `reps` times with an event after every
    replay and no host synchronisation; the figure is the MEDIAN over the replays of (with - without) / steps.  It
    contains the launch's boundaries (the gap in front of it and the drain of its output), so it sits 1-3 us above
    rocprofv3's per-kernel duration of the same launch (profiles/rNN_*_kernel_stats.csv), and the nine figures add up
    to the step.  Why not n launches of one kernel back to back: under a sustained run of one GEMM the chip clocks
    down (33 -> 44 us over 250 launches of the paired kernel), and an event pair around a launch inside the step costs
    a ~6 us bubble -- neither is the kernel's duration in the step.
    Returns a list of dicts (one per launch, issue order)."""
    import torch
    from rawaudiovae_kelsey_amd import engine as E
    from rawaudiovae_kelsey_amd._lib import lib
    ts = torch.cuda.current_stream()
    rows = step_launches(eng)

    def capture(mask):
        lib().rv_plan_diag_skip(eng._plan, mask)
        try:
            g = E.Graph(ts)
            with g:
                for _ in range(steps):
                    eng.step(x, stream=ts)
        finally:
            lib().rv_plan_diag_skip(eng._plan, 0)
        eng.host_steps -= steps
        return g
    eng.step(x, stream=ts)
    ts.synchronize()
    full = capture(0)
    full.launch()
    out = []
    for k, (name, flops, nbytes) in enumerate(rows):
        part = capture(1 << k)
        part.launch()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps + 1)]   # HIP events
        ev[0].record(ts)
        for r in range(reps):
            full.launch()
            ev[2 * r + 1].record(ts)
            part.launch()
            ev[2 * r + 2].record(ts)
        ev[-1].synchronize()
        d = sorted((ev[2 * r].elapsed_time(ev[2 * r + 1]) - ev[2 * r + 1].elapsed_time(ev[2 * r + 2])) / steps for r in range(reps))
        us = d[len(d) // 2] * 1e3
        row = {"launch": k, "kernel": name, "us": us, "us_min": d[0] * 1e3, "us_max": d[-1] * 1e3}
        if flops:
            row["tflops"] = flops / (us * 1e-6) / 1e12
            row["mfma_frac"] = row["tflops"] / PEAK_BF16_TFLOPS
        row["hbm_gbs"] = nbytes / (us * 1e-6) / 1e9
        row["hbm_frac"] = row["hbm_gbs"] / PEAK_HBM_GBS
        row["algorithmic_flops"], row["algorithmic_bytes"] = flops, nbytes
        out.append(row)
        del part
    # one more full-vs-full pair: the noise floor of the method
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record(ts); full.launch(); ev[1].record(ts); full.launch(); ev[2].record(ts)
    ev[2].synchronize()
    return out, abs(ev[0].elapsed_time(ev[1]) - ev[1].elapsed_time(ev[2])) / steps * 1e3


def time_api_loop(dev, pool, steps):
    """Side line `alt_api_loop` (never the headline): the reference's training loop UNCHANGED (train.py:184-193) through
    the drop-in surface -- `rawvae.model.VAE`, `loss_function`, `torch.optim.Adam(model.parameters())` -- on the headline's
    shape and frames.  Host-bound (two autograd nodes and the stock optimizer wrapper per step), so the figure is the
    host's as much as the GPU's; the same loop with the optimizer hook off (stock foreach Adam) and with loss_function on
    the general autograd route (`fused_loss = False`: round 4's loop, two autograd nodes) are timed beside it."""
    import torch
    from rawaudiovae_kelsey_amd import optim_hook
    from rawaudiovae_kelsey_amd.synth import make_params
    from rawvae.model import VAE, loss_function
    out = {"what": "the reference loop unchanged (zero_grad / model(x) / loss_function / backward / torch.optim.Adam.step) "
                   "through rawvae.model at C2; host-bound"}
    try:
        def run(hook, fused_loss=True):
            optim_hook.enabled = hook
            torch.manual_seed(0)
            m = VAE(S, H, L)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(S, H, L, 0).items()})
            m = m.to(dev)
            m.fused_loss = fused_loss
            opt = torch.optim.Adam(m.parameters(), lr=LR)

            def step(x):
                opt.zero_grad()
                recon, mu, logvar = m(x)
                loss = loss_function(recon, x, mu, logvar, KL_BETA, S)
                loss.backward()
                opt.step()
                return loss
            for i in range(10):
                step(pool[i % len(pool)])
            torch.cuda.synchronize()
            reps = []
            for r in range(5):
                t0 = time.perf_counter()
                for i in range(steps):
                    loss = step(pool[i % len(pool)])
                torch.cuda.synchronize()
                reps.append((time.perf_counter() - t0) / steps)
            reps.sort()
            return reps[len(reps) // 2], float(loss.item())
        n0 = optim_hook.stats["fused_steps"]
        t_on, loss_on = run(True)
        took = optim_hook.stats["fused_steps"] - n0
        t_off, _ = run(False)
        t_two, _ = run(True, fused_loss=False)
        out.update({"ms_per_step": t_on * 1e3, "value": float(B) / t_on, "unit": "frames/s", "final_loss": loss_on,
                    "optimizer_steps_taken_by_the_fused_kernel": took, "repeats": 5,
                    "ms_per_step_stock_optimizer_step": t_off * 1e3,
                    "ms_per_step_loss_on_the_general_autograd_route": t_two * 1e3})
    except Exception as exc:   # the headline is already measured: report, do not lose it
        out["error"] = str(exc)[:200]
    finally:
        optim_hook.enabled = True
    return out


def time_deep_c4(dev, comp, steps, warmup):
    """Side line `alt_deep_c4` (BASELINE configs[3], never the headline): the deep variant's training step at
    S=2048, H=2048, L=256, three H x H layers per side, B=4096, bf16 -- eager launches on `comp`, median of >= 5
    passes -- and its dominant kernel (the paired backward launch of one H x H layer: dX = relu'(dY W) 4096x2048x2048
    + dW = dY^T X, 68.7 GFLOP) timed live with HIP events on that stream, on the engine's own operands."""
    import torch
    from rawaudiovae_kelsey_amd.deep import DeepVAE
    from rawaudiovae_kelsey_amd._lib import lib, ptr
    Sd, Hd, Ld, depth, Bd = 2048, 2048, 256, 3, 4096
    torch.manual_seed(0)
    m = DeepVAE(Sd, Hd, Ld, depth).to(dev)
    eng = m.engine(Bd, kl_beta=KL_BETA, lr=LR, seed=0)
    g = torch.Generator(device=dev).manual_seed(99)
    xs = [torch.rand(Bd, Sd, device=dev, generator=g) * 2 - 1 for _ in range(4)]
    with torch.cuda.stream(comp):
        for i in range(warmup):
            eng.step(xs[i % 4], stream=comp)
        torch.cuda.synchronize()
        reps = []
        for r in range(5):
            t0 = time.perf_counter()
            for i in range(steps):
                eng.step(xs[i % 4], stream=comp)
            torch.cuda.synchronize()
            reps.append(time.perf_counter() - t0)
        reps.sort()
        med = reps[len(reps) // 2]
        # dominant kernel: backward of dec.1 (dy = d_dec[2] [Bp,Hp], W = dec.2.weight [Hp,Hp], x = dec_act[1])
        Lb = lib()
        Bp, Hp = eng.Bp, eng.Hp
        wname = "dec.%d.weight" % (depth - 1)
        st = comp.cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # HIP events on `comp`

        def launch():
            Lb.rv_linear_dgrad_wgrad(ptr(eng.d_dec[depth - 1]), Hp, ptr(eng.shadow[wname]), Hp, ptr(eng.dec_act[depth - 2]), Hp,
                                     Bp, Hp, Hp, ptr(eng.d_dec[depth - 2]), Hp, ptr(eng.bias_part["dec.%d.bias" % (depth - 2)]),
                                     ptr(eng.slabs[wname]), Hp, eng.splits[wname], *eng._slab_args(wname), st)
        for _ in range(5):
            launch()
        e0.record(comp)
        for _ in range(30):
            launch()
        e1.record(comp)
        e1.synchronize()
        kern_us = e0.elapsed_time(e1) / 30 * 1e3
    w = Sd * Hd + (depth - 1) * Hd * Hd + 2 * Hd * Ld + Ld * Hd + (depth - 1) * Hd * Hd + Hd * Sd
    fpf = 6 * w - 2 * Sd * Hd            # fwd + dgrad + wgrad per weight; the first layer has no dgrad
    kern_flops = 4.0 * Bd * Hd * Hd
    ach = kern_flops / (kern_us * 1e-6) / 1e12
    return {"what": "deep variant (BASELINE configs[3]): S=2048 H=2048 L=256, 3 H x H layers per side, B=4096, bf16, "
                    "%s split-K slabs for the large weight gradients, eager launches" % eng.slab_dtype,
            "ms_per_step": med / steps * 1e3, "value": float(Bd) * steps / med, "unit": "frames/s",
            "step_tflops": float(Bd) * steps / med * fpf / 1e12,
            "step_mfma_frac": float(Bd) * steps / med * fpf / 1e12 / PEAK_BF16_TFLOPS,
            "final_loss": eng.last_loss()[0], "repeats": len(reps),
            "roofline": {"bound": "mfma", "kernel": "gemm_dgrad_wgrad_kernel<256,256> (backward of one 2048 x 2048 layer, one "
                         "launch: dX=relu'(dY W) 4096x2048x2048 + dW=dY^T X split-K %d)" % eng.splits[wname],
                         "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                         "us_per_launch": kern_us}}


def host_cpu_quota():
    """CPUs this process may actually use: min(affinity mask, cgroup v2 / v1 CPU quota).  A one-GPU box is a slice of a
    256-thread host whose affinity mask still shows every CPU; the quota is what the slice is."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:           # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = float(f.read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world),
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)

    import torch
    import torch.distributed as dist
    from rawaudiovae_kelsey_amd.synth import flops_per_frame, make_frames, make_params
    # train.py refuses to run on an unhealthy collective stream (ddp.pick_comm_stream raises, on every rank together); a
    # benchmark must still produce its line: it runs, and `comm_stream_pick` in the JSON shows the round trip measured
    os.environ.setdefault("RV_COMM_STREAM_ALLOW_SLOW", "1")
    from rawaudiovae_kelsey_amd import ddp
    from rawaudiovae_kelsey_amd import engine as E

    # one process per GPU; the modulo only matters for the 2-rank plumbing rehearsal on a one-GPU box
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("RV_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" for rehearsals
    force_ddp = os.environ.get("RV_FORCE_DDP") == "1"    # exercise the data-parallel step on one rank
    if world > 1 or (force_ddp and "MASTER_ADDR" in os.environ and "RANK" in os.environ):
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    multi = dist.is_initialized()

    def agree(ok, what, reason=""):
        """Every rank reports whether its part of section `what` succeeded; the answer (all succeeded?) is the same on
        every rank.  Runs over torch.distributed, which is separate from the library-driven communicator."""
        if not multi:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        good = bool(int(flag.item()))
        if not ok:
            print("bench.py rank %d: %s failed: %s" % (rank, what, reason), file=sys.stderr, flush=True)
        return good

    def die_together(what, reason):
        """A failure INSIDE a data-parallel step (an exception between two collectives) leaves this rank's peers inside a
        collective it will never join: nothing after that point can be trusted, so every rank leaves, non-zero."""
        print("bench.py rank %d: %s: %s -- aborting on every rank" % (rank, what, reason), file=sys.stderr, flush=True)
        os._exit(5)

    ekw = {"slab_dtype": args.slab_dtype} if args.slab_dtype else {}
    eng = E.TrainEngine(S, H, L, B, device=dev, kl_beta=KL_BETA, lr=LR, seed=1000 + rank, ring=256, **ekw)
    eng.load_params(make_params(S, H, L, 0))
    if args.latent_fused is not None:
        eng.set_latent_fused(args.latent_fused)
    if args.roctx:
        eng.set_roctx(True)
    pool = [torch.from_numpy(make_frames(B, S, 1234 + 100 * rank + i)).to(dev) for i in range(POOL)]
    comp = torch.cuda.Stream(device=dev)
    use_ddp = world > 1 or force_ddp
    use_graph = not use_ddp and args.graph and not args.no_graph

    # ---- data-parallel step.  Default: the library issues the RCCL collectives itself (one host call per step:
    # ddp.NativeDdpRunner / rv_plan_step_ddp).  RV_DDP=torch (or a non-RCCL rehearsal backend, or an RCCL communicator
    # that cannot be created) selects the torch.distributed route: six host calls + three dist.all_reduce per step.
    runner, ddp_mode, comm = None, None, None
    native_fallback_reason, startup_check = None, None
    # the bench's exchange is an explicit choice, named in the line (config.ddp_payload): bf16 gradient payload -- what the
    # 8-GPU target is sized against -- with the library's default (fp32, the exact mean) timed beside it (alt_fp32_payload)
    payload = os.environ.get("RV_DDP_PAYLOAD", ddp.BENCH_PAYLOAD)
    rehearsal = False
    if use_ddp:
        # RV_DDP_REHEARSAL=shm (with RV_DIST_BACKEND=gloo): the library-driven step through the functional stand-in
        # collectives of tools/fake_collective.hip -- every line of this N > 1 branch on a ONE-GPU box, where RCCL itself
        # refuses two ranks on a device.  A rehearsal of the flow, never a measurement: the line says so.
        rehearsal = multi and backend != "nccl" and os.environ.get("RV_DDP_REHEARSAL") == "shm"
        want_native = multi and (backend == "nccl" or rehearsal) and os.environ.get("RV_DDP", "native") != "torch"
        ok = False
        if not want_native:
            native_fallback_reason = "RV_DDP=torch" if os.environ.get("RV_DDP") == "torch" else \
                ("backend %s is not RCCL" % backend if multi else "no process group")
        else:
            why = ""
            try:
                if rehearsal:
                    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
                    import standin_comm
                    comm = standin_comm.ShmComm(standin_comm.load(), "/rv_bench_%s" % os.environ.get("MASTER_PORT", "0"),
                                                world, rank, (eng.n_params + 8192) * 4)
                else:
                    comm = ddp.RcclComm()
                comm.self_test(dev)
                ok = True
            except Exception as exc:
                why = "%s: %s" % (type(exc).__name__, str(exc)[:300])
            # creating the communicator is collective (ncclCommInitRank): a rank whose creation FAILED returned from it,
            # and so did its peers -- a clean state, every rank can fall back together
            ok = agree(ok, "RCCL communicator / self-test", why)
            if not ok:
                reasons = [None] * world
                dist.all_gather_object(reasons, why)
                native_fallback_reason = "; ".join("rank %d: %s" % (i, r) for i, r in enumerate(reasons) if r) or "another rank failed"
        if ok:
            # The library-driven step is checked before it is timed, on scratch engines: three steps of it against three
            # steps of the torch.distributed route from the same weights on the same batches.  Replicas must end
            # bit-identical, and the two routes' parameters must agree (Adam moves every element by ~lr per step, so
            # agreement is a mean |difference| far below lr; the bf16 payload rounds each rank's gradient once, which
            # flips the sign of some near-zero gradients: those elements differ by 2 lr).
            def checked():
                ea = E.TrainEngine(S, H, L, B, device=dev, kl_beta=KL_BETA, lr=LR, seed=7, ring=16)
                eb = E.TrainEngine(S, H, L, B, device=dev, kl_beta=KL_BETA, lr=LR, seed=7, ring=16)
                for e in (ea, eb):
                    e.load_params(make_params(S, H, L, 0))
                # (the checked runner defers its tail exactly as the timed one will: RV_DDP_DEFER)
                ra = ddp.NativeDdpRunner(ea, comm, comp, payload=payload, defer=os.environ.get("RV_DDP_DEFER", "1") == "1")
                rb = ddp.DdpRunner(eb, ddp.GradSync(eb.grad, ddp.engine_buckets(eb)), comp, use_graphs=False)
                with torch.cuda.stream(comp):   # same seed and step counters: both engines draw the same eps
                    for i in range(3):
                        ra.step(pool[i % POOL])
                        rb.step(pool[i % POOL])
                    ra.flush()
                torch.cuda.synchronize()
                chk = torch.stack([ea.param.double().sum(), ea.param.double().abs().sum(),
                                   ea.buffer("W4b", torch.bfloat16, (-1,)).double().abs().sum()])
                lo, hi = chk.clone(), chk.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                same = bool(torch.equal(lo, hi))
                worst = torch.tensor([float((ea.param - eb.param).abs().mean())], dtype=torch.float64, device=dev)
                dist.all_reduce(worst, op=dist.ReduceOp.MAX)
                diff = float(worst.item())
                return same and diff < 0.05 * LR, "replicas %s, mean |param - torch.distributed route| = %.3g (lr %.1g)" % (
                    "identical" if same else "DIVERGED", diff, LR)
            try:
                good, startup_check = checked()
            except Exception as exc:   # between two collectives of a step: the peers cannot be reached any more
                die_together("startup check of the library-driven step raised", "%s: %s" % (type(exc).__name__, str(exc)[:300]))
            # a completed check that found a mismatch is a clean state (every collective of it has completed):
            # every rank saw the same verdict (it is made of all-reduced numbers) and falls back together
            if not agree(good, "startup check", startup_check):
                native_fallback_reason = "library-driven step failed the startup check (%s)" % startup_check
                ok = False
        if ok:
            # RV_DDP_DEFER=0: every step completes itself (default 1: a step's last wait + update go out behind the next
            # step's cast launch -- include/rawvae_hip.h RV_OPT_DDP_DEFER_TAIL; the timed region ends with the flush)
            runner = ddp.NativeDdpRunner(eng, comm, comp, payload=payload,
                                         defer=os.environ.get("RV_DDP_DEFER", "1") == "1")
            ddp_mode = ("%s all-reduce, 2 buckets (fc4 | fc1,fc21,fc22,fc3) issued by rv_plan_step_ddp on its collective "
                        "stream: fc4's behind the rest of backward, the second behind Adam(fc4)%s" % (
                            runner.payload, "; a step's last wait + update enqueued behind the next step's cast" if runner.defer else ""))
            ddp_mode += ", hipGraph" if runner.use_graph else ""
        else:
            sync = ddp.GradSync(eng.grad, ddp.engine_buckets(eng))
            runner = ddp.DdpRunner(eng, sync, comp, use_graphs=False)
            ddp_mode = "fp32 all-reduce, 3 buckets (fc4 | fc1 | rest) via torch.distributed, overlapped with backward"

    graphs = []
    with torch.cuda.stream(comp):
        # eager warm-up step (sets kernel attributes before any capture); with several ranks it must already be a
        # data-parallel step, or the replicas would start from different weights
        try:
            if use_ddp:
                runner.step(pool[0])
            else:
                eng.step(pool[0], stream=comp)
            torch.cuda.synchronize()
        except Exception as exc:
            if use_ddp and multi:
                die_together("first data-parallel step raised", "%s: %s" % (type(exc).__name__, str(exc)[:300]))
            raise
        pool_graph = None
        if use_graph:
            for x in pool:
                g = E.Graph(comp)
                with g:
                    eng.step(x, stream=comp)
                graphs.append(g)
        elif args.graph_pool and not use_ddp:
            pool_graph = E.Graph(comp)
            with pool_graph:
                for x in pool:
                    eng.step(x, stream=comp)
            eng.host_steps -= POOL   # the capture itself ran nothing

        def one_step(i):
            if use_ddp:
                runner.step(pool[i % POOL])
            elif use_graph:
                graphs[i % POOL].launch()
                eng.host_steps += 1
            else:
                eng.step(pool[i % POOL], stream=comp)

        def timed_pass(step_fn, first):
            """EXACTLY K steps between barrier + synchronize on both sides; max over ranks."""
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if pool_graph is not None and step_fn is one_step:
                for _ in range(args.steps // POOL):     # 8 steps per replay, in pool order
                    pool_graph.launch()
                eng.host_steps += (args.steps // POOL) * POOL
                for i in range(args.steps % POOL):
                    step_fn(i)
            else:
                for i in range(args.steps):
                    step_fn(first + i)
            if use_ddp and hasattr(runner, "flush"):
                runner.flush()                # the K-th step's deferred update belongs to the K steps
            host = time.perf_counter() - t0   # all K steps enqueued (the host runs ahead of the GPU)
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if multi:
                t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return el, host

        def timed_passes(step_fn, n_rep=None, cap=2000):
            """Median of >= 5 passes and >= --min-seconds of timed work in total; every rank runs the same
            number of passes (the count comes from rank 0's first pass)."""
            first_dt, host_dt = timed_pass(step_fn, args.warmup)
            if n_rep is None:
                n_rep = args.repeats if args.repeats > 0 else max(5, int(args.min_seconds / max(first_dt, 1e-6)) + 1)
                n_rep = min(n_rep, cap)
                if multi:
                    t = torch.tensor([n_rep], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
                    dist.broadcast(t, 0)
                    n_rep = int(t.item())
            passes = [first_dt]
            for r in range(1, n_rep):
                passes.append(timed_pass(step_fn, args.warmup + r * args.steps)[0])
            passes.sort()
            return passes, host_dt

        def replicas_identical(e):
            """Replicas must hold identical weights after identical averaged updates."""
            if not multi or world == 1:
                return None
            chk = torch.stack([e.param.double().sum(), e.param.double().abs().sum(),
                               e.buffer("W1b", torch.bfloat16, (-1,)).double().sum(),
                               e.buffer("W4b", torch.bfloat16, (-1,)).double().abs().sum()])
            if backend != "nccl":
                chk = chk.cpu()
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            same = bool(torch.equal(lo, hi))
            if rank == 0 and not same:
                print("bench.py: replicas diverged: %r vs %r" % (lo.tolist(), hi.tolist()), file=sys.stderr)
            return same

        try:
            for i in range(args.warmup):
                one_step(i)
            torch.cuda.synchronize()
            passes, host_dt = timed_passes(one_step)
        except Exception as exc:
            if use_ddp and multi:
                die_together("a timed data-parallel step raised", "%s: %s" % (type(exc).__name__, str(exc)[:300]))
            raise
        dt, dt_min, dt_max = passes[len(passes) // 2], passes[0], passes[-1]
        last = eng.losses(min(8, args.steps))
        replicas_consistent = replicas_identical(eng)

        # ---- side lines at N > 1 (never the headline; each is entered by every rank or by none) ----
        ddp_alts = {}
        if isinstance(runner, ddp.NativeDdpRunner) and os.environ.get("RV_DDP_ALT", "1") != "0" \
                and agree(replicas_consistent is not False, "headline replicas identical"):
            # the same schedule with the OTHER gradient payload (fp32 = the exact mean, twice the bytes on the links),
            # on the same engine: what the exchange costs at this GPU count
            other = "fp32" if runner.payload == "bf16" else "bf16"
            try:
                runner.set_payload(other)
                for i in range(5):
                    one_step(i)
                torch.cuda.synchronize()
                ap, _ = timed_passes(one_step, n_rep=min(len(passes), 7))
                ddp_alts["alt_%s_payload" % other] = {
                    "grad_allreduce": "%s payload, same schedule" % other, "ms_per_step": ap[len(ap) // 2] / args.steps * 1e3,
                    "value": float(B) * world * args.steps / ap[len(ap) // 2], "repeats": len(ap)}
                runner.set_payload("bf16" if other == "fp32" else "fp32")
            except Exception as exc:
                if multi:
                    die_together("side line (other payload) raised", "%s: %s" % (type(exc).__name__, str(exc)[:300]))
                ddp_alts["alt_%s_payload" % other] = {"error": str(exc)[:200]}
        if isinstance(runner, ddp.NativeDdpRunner) and runner.defer and os.environ.get("RV_DDP_ALT", "1") != "0" \
                and agree(replicas_consistent is not False, "headline replicas identical (before alt_no_defer)"):
            # the same schedule and payload with every step completing itself (RV_OPT_DDP_DEFER_TAIL off: the library's and
            # train.py's default) on a second engine -- round-5 advisor: the headline's deferred tail was sized on a one-GPU
            # model; this is its measured worth on the node the line was taken on
            try:
                eng2 = E.TrainEngine(S, H, L, B, device=dev, kl_beta=KL_BETA, lr=LR, seed=1000 + rank, ring=256)
                eng2.load_params(make_params(S, H, L, 0))
                run2 = ddp.NativeDdpRunner(eng2, comm, comp, payload=runner.payload, defer=False)
                for i in range(args.warmup + 1):
                    run2.step(pool[i % POOL])
                torch.cuda.synchronize()
                sp_, _ = timed_passes(lambda i: run2.step(pool[i % POOL]), n_rep=min(len(passes), 7))
                ddp_alts["alt_no_defer"] = {"grad_allreduce": "%s payload, every step completing itself" % run2.payload,
                                            "ms_per_step": sp_[len(sp_) // 2] / args.steps * 1e3,
                                            "value": float(B) * world * args.steps / sp_[len(sp_) // 2], "repeats": len(sp_)}
                rc2 = replicas_identical(eng2)
                if rc2 is not None:
                    ddp_alts["alt_no_defer"]["replicas_consistent"] = rc2
                del run2, eng2
            except Exception as exc:
                if multi:
                    die_together("side line (no deferred tail) raised", "%s: %s" % (type(exc).__name__, str(exc)[:300]))
                ddp_alts["alt_no_defer"] = {"error": str(exc)[:200]}

        # ---- per-launch timing and side lines at N = 1 ----
        launches, noise_us = None, None
        if rank == 0 and not use_ddp and not args.step_kernels_only:
            launches, noise_us = time_launches_in_step(eng, pool[0])
        alts = {}
        if not use_ddp and not args.no_alts and not args.step_kernels_only:
            def time_alt(what, **kw):
                try:
                    e2 = E.TrainEngine(S, H, L, B, device=dev, kl_beta=KL_BETA, lr=LR, seed=1000 + rank, ring=256, **kw)
                    e2.load_params(make_params(S, H, L, 0))
                    for i in range(args.warmup + 5):
                        e2.step(pool[i % POOL], stream=comp)
                    torch.cuda.synchronize()
                    reps = []
                    for r in range(max(5, min(len(passes), 50))):
                        t0 = time.perf_counter()
                        for i in range(args.steps):
                            e2.step(pool[(r * args.steps + i) % POOL], stream=comp)
                        torch.cuda.synchronize()
                        reps.append(time.perf_counter() - t0)
                    reps.sort()
                    med = reps[len(reps) // 2]
                    return {"what": what, "ms_per_step": med / args.steps * 1e3, "value": float(B) * args.steps / med,
                            "final_loss": e2.losses(1)[-1], "repeats": len(reps)}
                except Exception as exc:   # the headline is already measured: report, do not lose it
                    return {"what": what, "error": str(exc)[:200]}
            def time_shape(what, S2, H2, L2, B2):
                """The training step at another model / batch shape (the reference's own .ini files), timed as the headline
                is (eager launches, median over passes) with every launch group timed in the step (`kernels`)."""
                try:
                    torch.cuda.empty_cache()
                    n_pool = 2 if B2 > 16384 else POOL
                    steps2 = max(5, args.steps * 4096 // max(B2, 4096))
                    pool2 = [torch.from_numpy(make_frames(B2, S2, 7000 + i)).to(dev) for i in range(n_pool)]
                    e2 = E.TrainEngine(S2, H2, L2, B2, device=dev, kl_beta=KL_BETA, lr=LR, seed=1000 + rank, ring=256)
                    e2.load_params(make_params(S2, H2, L2, 0))
                    for i in range(max(3, min(args.warmup, steps2))):
                        e2.step(pool2[i % n_pool], stream=comp)
                    torch.cuda.synchronize()
                    reps = []
                    for r in range(5):
                        t0 = time.perf_counter()
                        for i in range(steps2):
                            e2.step(pool2[(r * steps2 + i) % n_pool], stream=comp)
                        torch.cuda.synchronize()
                        reps.append(time.perf_counter() - t0)
                    reps.sort()
                    med = reps[len(reps) // 2]
                    fps = float(B2) * steps2 / med
                    rows, noise = time_launches_in_step(e2, pool2[0], steps=max(1, 10 * 4096 // max(B2, 4096)))
                    out2 = {"what": what, "workload": "S=%d H=%d L=%d, per-GPU batch %d" % (S2, H2, L2, B2),
                            "ms_per_step": med / steps2 * 1e3, "value": fps, "steps": steps2, "repeats": len(reps),
                            "flops_per_frame": flops_per_frame(S2, H2, L2),
                            "step_tflops": fps * flops_per_frame(S2, H2, L2) / 1e12,
                            "step_mfma_frac": fps * flops_per_frame(S2, H2, L2) / 1e12 / PEAK_BF16_TFLOPS,
                            "final_loss": e2.losses(1)[-1], "sum_of_launches_us": sum(r_["us"] for r_ in rows),
                            "method_noise_us": noise,
                            "kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r_.items()} for r_ in rows]}
                    del e2, pool2
                    torch.cuda.empty_cache()
                    return out2
                except Exception as exc:
                    return {"what": what, "error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
            # the reference's own configurations (never the headline: BASELINE.json quotes the metric on configs[1] = C2)
            alts["alt_ref_ini"] = time_shape(
                "the reference's own model -- segment_length 1024, n_units 2048, latent_dim 256 (default.ini:3,18-19; "
                "kelsey_iterable.ini:17-18) -- at the batch its HPC run used (kelsey_iterable.ini:26: 4096): the latent-sized "
                "launches are GEMMs with the reparameterisation in their epilogues (csrc/latent.hip)", 1024, 2048, 256, 4096)
            alts["alt_default_ini"] = time_shape(
                "default.ini as shipped (default.ini:3,18-19,27: the same model at batch_size 131072): 32 output tiles per CU, "
                "the large GEMMs on 256 x 256 tiles (csrc/gemm_launch.hip big_tiles)", 1024, 2048, 256, 131072)
            alts["alt_fp8"] = time_alt(
                "fp8 weight path (BASELINE configs[4]): all four large GEMM launches on e4m3 operands -- fc1 / fc4 forward, fc4's "
                "backward (dgrad + wgrad, one 256x256 launch) and fc1's weight gradient (256x256, beside the optimizer riders, which "
                "hand 15 % of their work to the GEMM blocks) -- v_mfma_scale_f32_16x16x128_f8f6f4, 128-deep K tiles, per-tensor "
                "scales, delayed scaling for h3 and dP1, fixed scale for dP4; heads and fc3 bf16", fp8=True)
            alts["alt_fp8_forward_only"] = time_alt("round 3's fp8 path: fc1 / fc4 forward on e4m3 operands, whole backward bf16", fp8="fwd")
            try:
                alts["alt_deep_c4"] = time_deep_c4(dev, comp, max(10, args.steps // 4), 5)
            except Exception as exc:
                alts["alt_deep_c4"] = {"what": "deep variant (BASELINE configs[3])", "error": str(exc)[:200]}
            alts["alt_fp32_slabs"] = time_alt(
                "split-K partial sums of dW1 / dW4 (and of the heads' weight gradients) stored as fp32 instead of "
                "block-floating-point fp16 (round 2's default); everything else as the headline", slab_dtype="fp32")
            alts["alt_api_loop"] = time_api_loop(dev, pool, max(50, min(args.steps, 200)))

    if not all(map(lambda v: v == v and abs(v) < 1e3, last)):
        print("bench.py: non-finite loss %r" % (last,), file=sys.stderr)
        sys.exit(3)

    if rank == 0:
        frames = float(B) * world * args.steps
        value = frames / dt
        # HBM bytes per launch of the longest kernel: NOT measured in this run -- the newest committed PMC summary
        # (profiles/rNN_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command,
        # corrected as MI355X_MICROARCH.md 'HBM' prescribes; tools/pmc_round.sh), labelled with its source
        roofline = None
        if launches:
            top = max(launches, key=lambda r: r["us"])
            traffic, traffic_src = None, None
            try:
                import glob
                cand = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]*_traffic.json")))
                with open(cand[-1]) as f:
                    tj = json.load(f)
                per = tj.get("per_launch_bytes", {})
                traffic = per.get(str(top["launch"]), tj.get("traffic_bytes") if top["launch"] == 4 else None)
                traffic_src = "profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command on the " \
                              "builder's box, commit %s: counters serialise dispatches, so never the timed run itself)" % (
                                  os.path.basename(cand[-1]), tj.get("commit", "not recorded"))
            except Exception:
                pass
            mfma_bound = top.get("mfma_frac", 0.0) >= top["hbm_frac"]
            roofline = {
                "bound": "mfma" if mfma_bound else "hbm", "kernel": top["kernel"], "launch": top["launch"],
                "achieved": top["tflops"] if mfma_bound else top["hbm_gbs"],
                "peak": PEAK_BF16_TFLOPS if mfma_bound else PEAK_HBM_GBS, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": top["mfma_frac"] if mfma_bound else top["hbm_frac"],
                # a mixed launch (GEMM blocks on half the chip, optimizer blocks on the other half) has two fractions
                "mfma_frac": top.get("mfma_frac"), "hbm_frac": top["hbm_frac"],
                "traffic": traffic, "traffic_source": traffic_src, "us_per_launch": top["us"],
                "us_per_launch_range": [top["us_min"], top["us_max"]],
                "timing": "in the step, this run: median over 7 alternating replays of (hipGraph of 10 full steps) - (the same "
                          "graph without this launch), / 10, HIP events after every replay on the launching stream; launch "
                          "boundaries included; `kernels` holds all nine launches measured the same way",
                "sum_of_launches_us": sum(r["us"] for r in launches), "method_noise_us": noise_us}
        ms = [p_ / args.steps * 1e3 for p_ in passes]
        out = {
            "metric": "audio frames/sec (fwd+bwd+step), 1024-sample frames",
            "value": value if replicas_consistent is not False else None,
            "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "C2 raw-audio VAE train step: S=1024 H=2048 L=64, per-GPU batch 4096, "
                                   "kl_beta=1e-4, Adam lr=1e-4", "global_batch": B * world,
                       "parallelism": "dp%d" % world,
                       "launch": "hipGraph (one graph of %d steps)" % POOL if pool_graph is not None else
                                 "hipGraph" if (use_graph or getattr(runner, "use_graph", False)) else "eager",
                       "wgrad_slabs": eng.slab_dtype,
                       **({"ddp_mode": "allreduce",
                           "ddp_payload": getattr(runner, "payload", "fp32"),
                           # how the payload is summed across ranks: by the collective library, in the payload's own type
                           # (RCCL's rings add bf16 hop by hop: world - 1 roundings per element; error model and gate in
                           # DESIGN.md section 5 and tests/ddp_shm_worker.py: 2.9e-3 rel-L2 of the averaged gradient against
                           # the oracle at world 4 in ring order, 2.2e-3 with one fp32 rounding, 7e-3 allowed)
                           "ddp_payload_accumulate": ("fp32 (exact mean)" if getattr(runner, "payload", "fp32") == "fp32" else
                                                      "bf16 hop by hop inside the collective (RCCL); stand-in rehearsals: fp32, "
                                                      "rounded once" if not rehearsal else "fp32, rounded once (stand-in collectives)")}
                          if runner is not None else {}),
                       "grad_allreduce": ddp_mode},
            # `value` / `ms_per_step` are the MEDIAN over `repeats` passes of exactly `steps` steps each
            "timing": {"repeats": len(passes), "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
                       "timed_seconds_total": sum(passes)},
            "host_us_per_step": host_dt / args.steps * 1e6,
            **ddp_alts, **alts,
            **({"replicas_consistent": replicas_consistent} if replicas_consistent is not None else {}),
            **({"native_fallback_reason": native_fallback_reason, "startup_check": startup_check,
                "rccl_version": getattr(comm, "version", None), "rccl_ranks": getattr(comm, "rccl_count", None),
                **({"rehearsal": "stand-in collectives through shared memory (tools/fake_collective.hip): the flow, not a measurement"}
                   if rehearsal else {}),
                "comm_stream_pick": [{"us_per_round_trip": u, "candidates_tried": n} for u, n in ddp.comm_stream_report()]}
               if use_ddp else {}),
            "step_tflops": value * flops_per_frame(S, H, L) / 1e12,
            "step_mfma_frac": value * flops_per_frame(S, H, L) / 1e12 / (PEAK_BF16_TFLOPS * world),
            # algorithmic HBM bytes per step: 54,784 B/frame of activations + 38 B/param
            "step_hbm_frac": (value * 54784.0 + (value / B / world) * 38.0 * 4592768 * world) / (8.0e12 * world),
            "final_loss": last[-1],
            "roofline": roofline,
            **({"kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in launches]}
               if launches else {}),
        }
        if world == 1 and not use_ddp and not args.no_cpu_baseline:
            from oracle.torch_port import cpu_description, time_cpu_step
            usable, quota = host_cpu_quota()
            threads = max(1, min(usable, int(os.environ.get("RV_CPU_THREADS", "16"))))
            fps, ms_c, n, threads = time_cpu_step(S, H, L, B, make_params(S, H, L, 0), make_frames(B, S, 1234),
                                                  seconds=args.cpu_seconds, threads=threads)
            # BASELINE configs[0] (the reference's own CPU-runnable case: 512-sample frames, latent 8, batch 32)
            s_fps, s_ms, s_n, _ = time_cpu_step(512, H, 8, 32, make_params(512, H, 8, 0), make_frames(32, 512, 1234),
                                                seconds=min(3.0, args.cpu_seconds), threads=threads)
            # SURVEY 8d's third CPU shape: the reference's own model (latent_dim 256) at batch 4096
            r_fps, r_ms, r_n, _ = time_cpu_step(1024, H, 256, 4096, make_params(1024, H, 256, 0), make_frames(4096, 1024, 1234),
                                                seconds=min(6.0, args.cpu_seconds), threads=threads)
            out["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": threads, "kind": "port",
                                   "sample": "%d steps of the same C2 step (B=4096) in stock PyTorch fp32 on the "
                                             "host, median %.1f ms/step" % (n, ms_c),
                                   "smoke_shape": {"value": s_fps, "unit": "frames/s", "ms_per_step": s_ms,
                                                   "sample": "%d steps of S=512 H=2048 L=8 B=32" % s_n},
                                   "ref_ini_shape": {"value": r_fps, "unit": "frames/s", "ms_per_step": r_ms,
                                                     "sample": "%d steps of S=1024 H=2048 L=256 B=4096 (the model of default.ini / "
                                                               "kelsey_iterable.ini at the latter's batch)" % r_n},
                                   "cgroup_cpu_quota": quota, "cpus_after_quota": usable,
                                   **cpu_description()}
        print(json.dumps(out))
    code = 4 if replicas_consistent is False else 0   # replicas with different weights are not a result
    if multi:
        dist.barrier()   # rank 0 did extra timing work; leave together
        if comm is not None:
            torch.cuda.synchronize()
            comm.destroy()
        dist.destroy_process_group()
        if world > 1:
            # every rank's verdict is in `code` and the line is printed: leave without interpreter teardown (the unload
            # order of torch, RCCL, the HIP runtime and ctypes-loaded libraries at exit must not decide a rank's exit
            # status).  Not in the one-rank rehearsal: a profiler wrapped around it writes its output at exit.
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(code)
    if code:
        sys.exit(code)


if __name__ == "__main__":
    main()
