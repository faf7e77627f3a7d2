#!/usr/bin/env python3
"""Benchmark of the MI355X training path: audio frames/sec for a full training
step (forward + fused loss + backward + Adam) on synthetic 1024-sample frames.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] / configs[2]): S=1024, H=2048, L=64, per-GPU batch
4096, bf16 MFMA inputs with fp32 accumulation, fp32 master weights and Adam state.
A step is one pass of the hot path over one resident batch (a pool of 8 distinct
device-resident batches is cycled); eps is drawn on-device: one host call per step
(`rv_plan_step`) enqueues its 12 kernels back to back on one stream (`--graph` replays them from a
hipGraph instead: the kernels run as fast, but consecutive replays are ~8 us apart where eager launches
are back to back -- profiles/r02_graph_vs_eager_timeline.txt).  One process per GPU; with N > 1 each
step is one `rv_plan_step_ddp` call that also issues the RCCL collectives as backward produces the
gradients.  Two schedules exist (sharded optimizer = reduce-scatter, Adam on the local shard, all-gather; and
all-reduce + full Adam on every rank); both are timed with the same passes and the faster is the headline
(`ddp_schedule_pick`), the other is reported beside it.  Weak scaling: per-GPU batch fixed.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the paired fc4
backward GEMM launch, the longest kernel of the step), timed live with HIP events; `cpu_baseline`
is the stock-PyTorch CPU port of the same step (oracle/torch_port.py) timed on this
node's host cores (N=1 only).
"""
import argparse
import ctypes as C
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graph", action="store_true",
                    help="replay each step from a hipGraph (N=1); ~8 us/step slower than back-to-back eager launches "
                         "of the same 12 kernels (gap between replays), so eager is the default")
    ap.add_argument("--graph-pool", action="store_true",
                    help="(N=1) ONE hipGraph holding the steps of the whole batch pool (8 steps per replay), so the gap "
                         "between replays is paid once per 8 steps; steps beyond a multiple of 8 run eagerly")
    ap.add_argument("--no-graph", action="store_true", help="(default) eager launches")
    ap.add_argument("--latent-fused", type=int, default=None, choices=[0, 1],
                    help="heads + reparam + fc3 of the forward as one row-local launch (1, default) or three launches (0)")
    ap.add_argument("--fp8", action="store_true",
                    help="ignored (kept for old command lines): the headline is bf16; the fp8 forward is timed as the side line `alt_fp8`")
    ap.add_argument("--step-kernels-only", action="store_true",
                    help="profiling runs (tools/prof_round3.sh, tools/pmc_round.sh): launch nothing but the timed steps -- "
                         "the dominant kernel is then timed by ONE batch of 50 back-to-back launches instead of the in-step "
                         "differential measurement, whose phase-by-phase graphs would put other kernel variants into the "
                         "profile")
    ap.add_argument("--no-alts", action="store_true",
                    help="skip the side lines `alt_fp8` (fc1 / fc4 forward on e4m3 operands, BASELINE configs[4]) and "
                         "`alt_fp32_slabs`; they are timed after the headline at N=1 and never replace it")
    ap.add_argument("--slab-dtype", default=None, choices=["fp32", "fp16"],
                    help="element type of the fc1 / fc4 weight-gradient split-K slabs (default: the engine's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed passes of K steps each (0 = at least 5 and enough for --min-seconds of timed work)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="minimum total timed work (all repeats)")
    return ap.parse_args()


def time_dominant_kernel(eng, x, steps=10, reps=7):
    """Average duration of the step's longest kernel -- the paired fc4 backward
    (`gemm_dgrad_wgrad_kernel`: dP3 = relu'(dP4 W4) and dW4 = dP4^T h3 in one launch, 256x256 tiles)
    -- IN the step, with HIP events on the launching stream.  Two hipGraphs of `steps` training steps issued phase by
    phase through the training plan (`rv_plan_step`: forward | fc4 backward | rest of backward | Adam), one with and
    one without the fc4-backward phase (the plan's own launch: same operands, slab type and store policy as in the
    full step), are replayed alternately `reps` times with an event after every replay and no host synchronisation;
    the figure is the MEDIAN over the replays of (time with - time without) / steps.  It contains the kernel's
    boundaries (the launch gap and the drain of its 33 MB of output), so it sits 1-3 us above rocprofv3's per-kernel
    duration of the same launch (profiles/rNN_*_kernel_stats.csv).
    Why not `n` launches of the kernel back to back: under a sustained run of this one kernel (~1 PFLOP/s) the chip
    slows down -- 50-launch batches replayed from a graph went from 33-35 us (first) to 41-44 us (fifth), and issued
    from Python they are host-bound on top (a single-phase call costs ~38 us of host time) -- which is neither the
    kernel's duration in the step nor a property of the kernel.  Bracketing the launch inside the step with its own
    event pair is worse still: an event record in a busy stream costs a ~6 us bubble (measured: 37-41 us).
    Returns (ms_per_launch, algorithmic flops per launch, description, all replay differences in ms)."""
    import torch
    from rawaudiovae_kelsey_amd import engine as E
    from rawaudiovae_kelsey_amd._lib import dgrad_wgrad_pick
    Bp, Sp, Hp, Lp = eng.padded()
    paired, bm, splits = dgrad_wgrad_pick(Bp, Hp, Sp)   # what the training step itself uses
    ts = torch.cuda.current_stream()
    rest = E.PHASE_BWD_CHAIN | E.PHASE_BWD_REST
    seq_with = (E.PHASE_FWD, E.PHASE_BWD_FC4, rest, E.PHASE_ADAM)
    seq_without = (E.PHASE_FWD, rest, E.PHASE_ADAM)

    def run(seq):
        for ph in seq:
            eng.step(x, phases=ph, stream=ts)
    run(seq_with)
    ts.synchronize()
    graphs = []
    for seq in (seq_with, seq_without):
        g = E.Graph(ts)
        with g:
            for _ in range(steps):
                run(seq)
        g.launch()
        graphs.append(g)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps + 1)]   # HIP events
    ev[0].record(ts)
    for r in range(reps):
        graphs[0].launch()
        ev[2 * r + 1].record(ts)
        graphs[1].launch()
        ev[2 * r + 2].record(ts)
    ev[-1].synchronize()
    diffs = sorted((ev[2 * r].elapsed_time(ev[2 * r + 1]) - ev[2 * r + 1].elapsed_time(ev[2 * r + 2])) / steps for r in range(reps))
    desc = ("gemm_dgrad_wgrad_kernel<256,256> (fc4 backward, one launch: dX=relu'(dY W) 4096x2048x1024 + "
            "dW=dY^T X 1024x2048x4096 split-K %d, %s slabs)" % (splits, eng.slab_dtype)) if paired else \
        "rv_linear_dgrad + rv_linear_wgrad (fc4 backward, unpaired fallback, split-K %d)" % splits
    return diffs[len(diffs) // 2], 4.0 * S * H * B, desc, diffs


def time_dominant_kernel_batch(eng, x, reps=50):
    """`--step-kernels-only`: one pair of HIP events around `reps` back-to-back launches of the plan's fc4-backward
    phase (the round's earlier method; see time_dominant_kernel for why it is not the headline method)."""
    import torch
    from rawaudiovae_kelsey_amd import engine as E
    ts = torch.cuda.current_stream()
    eng.step(x, phases=E.PHASE_FWD, stream=ts)
    for _ in range(5):
        eng.step(x, phases=E.PHASE_BWD_FC4, stream=ts)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(ts)
    for _ in range(reps):
        eng.step(x, phases=E.PHASE_BWD_FC4, stream=ts)
    b.record(ts)
    b.synchronize()
    ms = a.elapsed_time(b) / reps
    return ms, 4.0 * S * H * B, "gemm_dgrad_wgrad_kernel<256,256> (fc4 backward; one batch of %d back-to-back launches)" % reps, [ms]


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
                                     ptr(eng.slabs[wname]), Hp, eng.splits[wname], 0, None, st)
        for _ in range(5):
            launch()
        e0.record(comp)
        for _ in range(30):
            launch()
        e1.record(comp)
        e1.synchronize()
        ms = C.c_float(e0.elapsed_time(e1))
    kern_us = ms.value / 30 * 1e3
    w = Sd * Hd + (depth - 1) * Hd * Hd + 2 * Hd * Ld + Ld * Hd + (depth - 1) * Hd * Hd + Hd * Sd
    fpf = 6 * w - 2 * Sd * Hd            # fwd + dgrad + wgrad per weight; the first layer has no dgrad
    kern_flops = 4.0 * Bd * Hd * Hd
    ach = kern_flops / (kern_us * 1e-6) / 1e12
    return {"what": "deep variant (BASELINE configs[3]): S=2048 H=2048 L=256, 3 H x H layers per side, B=4096, bf16, "
                    "fp32 split-K slabs, eager launches",
            "ms_per_step": med / steps * 1e3, "value": float(Bd) * steps / med, "unit": "frames/s",
            "step_tflops": float(Bd) * steps / med * fpf / 1e12,
            "step_mfma_frac": float(Bd) * steps / med * fpf / 1e12 / PEAK_BF16_TFLOPS,
            "final_loss": eng.last_loss()[0], "repeats": len(reps),
            "roofline": {"bound": "mfma", "kernel": "gemm_dgrad_wgrad_kernel<256,256> (backward of one 2048 x 2048 layer, one "
                         "launch: dX=relu'(dY W) 4096x2048x2048 + dW=dY^T X split-K %d)" % eng.splits[wname],
                         "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                         "us_per_launch": kern_us}}


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
    # train.py refuses to run on an unhealthy collective stream (ddp.pick_comm_stream raises); a benchmark must still
    # produce its line: it runs, and `comm_stream_pick` in the JSON shows the round trip that was measured
    os.environ.setdefault("RV_COMM_STREAM_ALLOW_SLOW", "1")
    from rawaudiovae_kelsey_amd import engine as E

    # one process per GPU; the modulo only matters for the 2-rank plumbing rehearsal on a one-GPU box
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("RV_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" for rehearsals
    if world > 1 or (os.environ.get("RV_FORCE_DDP") == "1" and "MASTER_ADDR" in os.environ and "RANK" in os.environ):
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    ekw = {"slab_dtype": args.slab_dtype} if args.slab_dtype else {}
    eng = E.TrainEngine(S, H, L, B, device=dev, kl_beta=KL_BETA, lr=LR, seed=1000 + rank, ring=256, **ekw)
    eng.load_params(make_params(S, H, L, 0))
    if args.latent_fused is not None:
        eng.set_latent_fused(args.latent_fused)
    pool = [torch.from_numpy(make_frames(B, S, 1234 + 100 * rank + i)).to(dev) for i in range(POOL)]
    comp = torch.cuda.Stream(device=dev)
    use_graph = world == 1 and args.graph and not args.no_graph and os.environ.get("RV_FORCE_DDP") != "1"
    from rawaudiovae_kelsey_amd import ddp
    force_ddp = os.environ.get("RV_FORCE_DDP") == "1"  # exercise the phased DDP step on one rank
    sync = ddp.GradSync(eng.grad, ddp.engine_buckets(eng)) if (world > 1 or force_ddp) else None

    # Data-parallel step.  Default: the library issues the RCCL all-reduces itself (one host call per
    # step, collectives on their own stream between the kernels: ddp.NativeDdpRunner / rv_plan_step_ddp).
    # RV_DDP=torch (or a non-RCCL rehearsal backend, or a failed RCCL self-test) selects the
    # torch.distributed path: six host calls + three dist.all_reduce per step (ddp.ddp_step).
    runner, ddp_mode, comm = None, None, None
    native_fallback_reason = None    # why the library-driven RCCL step was not used (None: it was, or N = 1)
    if sync is not None:
        want_native = dist.is_initialized() and backend == "nccl" and os.environ.get("RV_DDP", "native") != "torch"
        ok = 0
        if not want_native:
            native_fallback_reason = "RV_DDP=torch" if os.environ.get("RV_DDP") == "torch" else \
                "backend %s is not RCCL" % backend
        else:
            my_reason = ""
            try:
                comm = ddp.RcclComm()
                comm.self_test(dev)
                ok = 1
            except Exception as exc:  # fall back together, below
                my_reason = "rank %d: %s: %s" % (rank, type(exc).__name__, str(exc)[:300])
                print("bench.py: native RCCL path unavailable (%s)" % my_reason, file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
            if not ok:   # every rank's reason reaches rank 0's JSON line
                reasons = [None] * world
                dist.all_gather_object(reasons, my_reason)
                native_fallback_reason = "; ".join(r for r in reasons if r) or "another rank failed"
        # ONE default schedule for bench.py and train.py alike, fixed before anything is measured: all-reduce
        # (RV_DDP_MODE=sharded selects the sharded optimizer; the other schedule is timed as a side line)
        sharded = os.environ.get("RV_DDP_MODE", "allreduce") == "sharded"
        if ok and os.environ.get("RV_DDP_CHECK", "1") == "1":
            # The library-driven step has only ever run on ONE rank before this job (one-GPU development boxes), so
            # before it is timed it is checked here, on scratch engines: three steps of it against three steps of the
            # torch.distributed path from the same weights on the same batches.  Replicas must end bit-identical and
            # the two paths' parameters must agree (Adam moves every element by ~lr per step, so agreement is a
            # mean |difference| far below lr; elements whose tiny gradient changes sign between two summation
            # orders differ by 2 lr each).  A failing mode is dropped for the next one and the JSON says why.
            def checked(mode_sharded):
                ea = E.TrainEngine(S, H, L, B, device=dev, kl_beta=KL_BETA, lr=LR, seed=7, ring=16)
                eb = E.TrainEngine(S, H, L, B, device=dev, kl_beta=KL_BETA, lr=LR, seed=7, ring=16)
                for e in (ea, eb):
                    e.load_params(make_params(S, H, L, 0))
                ra = ddp.NativeDdpRunner(ea, comm, comp, sharded=mode_sharded)
                rb = ddp.DdpRunner(eb, ddp.GradSync(eb.grad, ddp.engine_buckets(eb)), comp, use_graphs=False)
                with torch.cuda.stream(comp):   # same seed and step counters: both engines draw the same eps
                    for i in range(3):
                        ra.step(pool[i % POOL])
                        rb.step(pool[i % POOL])
                torch.cuda.synchronize()
                ddp.gather_sharded_params(ea)
                chk = torch.stack([ea.param.double().sum(), ea.param.double().abs().sum(),
                                   ea.buffer("W4b", torch.bfloat16, (-1,)).double().abs().sum()])
                lo, hi = chk.clone(), chk.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                same = bool(torch.equal(lo, hi))
                diff = float((ea.param - eb.param).abs().mean())
                worst = torch.tensor([diff], dtype=torch.float64, device=dev)
                dist.all_reduce(worst, op=dist.ReduceOp.MAX)
                diff = float(worst.item())
                good = same and diff < 0.05 * LR
                return good, "replicas %s, mean |param - torch.distributed path| = %.3g (lr %.1g)" % (
                    "identical" if same else "DIVERGED", diff, LR)
            try:
                good, note = checked(sharded)
                if not good and sharded:
                    native_fallback_reason = "sharded step failed the startup check (%s); all-reduce schedule used" % note
                    sharded = False
                    good, note = checked(False)
                if not good:
                    native_fallback_reason = ((native_fallback_reason or "") +
                                              " native all-reduce step failed the startup check (%s)" % note).strip()
                    ok = 0
            except Exception as exc:
                native_fallback_reason = "startup check raised %s: %s" % (type(exc).__name__, str(exc)[:300])
                ok = 0
            flag = torch.tensor([ok, int(sharded)], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok, sharded = int(flag[0].item()), bool(int(flag[1].item()))
        if ok:
            # RV_DDP_MODE=sharded (default): sharded optimizer -- reduce-scatter gradients, Adam on 1/world of the
            # arena per rank, all-gather parameters; =allreduce: all-reduce + the full update on every rank
            runner = ddp.NativeDdpRunner(eng, comm, comp, use_graph=os.environ.get("RV_DDP_GRAPH") == "1", sharded=sharded)
            ddp_mode = ("sharded optimizer: fp32 reduce-scatter (fc4 | rest) -> Adam on this rank's 1/%d of the arena -> "
                        "all-gather of %s, all issued by rv_plan_step_ddp on its own stream"
                        % (world, "the 16-bit parameter message (bf16 weights + fp32 biases)"
                           if getattr(eng, "shard_gather", "fp32") == "bf16" else "the fp32 parameters")) \
                if sharded else ("fp32, 2 buckets (fc4 | fc1,fc21,fc22,fc3) issued by rv_plan_step_ddp on its own "
                                 "stream, overlapped with backward")
            ddp_mode += ", hipGraph" if runner.use_graph else ""
        else:
            # eager launches: six hipGraph segments per step measured slower (292 vs 263 us on one rank)
            runner = ddp.DdpRunner(eng, sync, comp, use_graphs=False)
            ddp_mode = "fp32, 3 buckets (fc4 | fc1 | rest) via torch.distributed, overlapped with backward"

    def ddp_step(x):
        runner.step(x)

    graphs = []
    with torch.cuda.stream(comp):
        # eager warm-up step (sets kernel attributes before any capture); with several ranks it must
        # already be a data-parallel step, or the replicas would start from different weights
        if sync is not None:
            runner.step(pool[0])
        else:
            eng.step(pool[0], stream=comp)
        torch.cuda.synchronize()
        pool_graph = None
        if use_graph:
            for x in pool:
                g = E.Graph(comp)
                with g:
                    eng.step(x, stream=comp)
                graphs.append(g)
        elif args.graph_pool and world == 1 and not force_ddp:
            pool_graph = E.Graph(comp)
            with pool_graph:
                for x in pool:
                    eng.step(x, stream=comp)
            eng.host_steps -= POOL   # the capture itself ran nothing

        def one_step(i):
            if sync is not None:
                ddp_step(pool[i % POOL])
            elif use_graph:
                graphs[i % POOL].launch()
                eng.host_steps += 1
            else:
                eng.step(pool[i % POOL], stream=comp)

        for i in range(args.warmup):
            one_step(i)
        torch.cuda.synchronize()

        def timed_pass(first):
            """EXACTLY K steps between barrier + synchronize on both sides; max over ranks."""
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if pool_graph is not None:
                for _ in range(args.steps // POOL):     # 8 steps per replay, in pool order
                    pool_graph.launch()
                eng.host_steps += (args.steps // POOL) * POOL
                for i in range(args.steps % POOL):
                    one_step(i)
            else:
                for i in range(args.steps):
                    one_step(first + i)
            host = time.perf_counter() - t0   # all K steps enqueued (the host runs ahead of the GPU)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return el, host

        # median of >= 5 passes and >= --min-seconds of timed work in total; every rank runs
        # the same number of passes (the count comes from rank 0's first pass)
        first_dt, host_dt = timed_pass(args.warmup)
        n_rep = args.repeats if args.repeats > 0 else max(5, int(args.min_seconds / max(first_dt, 1e-6)) + 1)
        n_rep = min(n_rep, 2000)
        if world > 1:
            t = torch.tensor([n_rep], dtype=torch.int64, device=dev)
            dist.broadcast(t, 0)
            n_rep = int(t.item())
        passes = [first_dt]
        for r in range(1, n_rep):
            el, _ = timed_pass(args.warmup + r * args.steps)
            passes.append(el)
        passes.sort()
        dt = passes[len(passes) // 2]
        dt_min, dt_max = passes[0], passes[-1]

        last = eng.losses(min(8, args.steps))
        replicas_consistent = None
        if world > 1:  # replicas must hold identical weights after identical averaged updates
            ddp.gather_sharded_params(eng)   # 16-bit parameter message: fp32 weight masters live on their owners
            chk = torch.stack([eng.param.double().sum(), eng.param.double().abs().sum(),
                               eng.buffer("W1b", torch.bfloat16, (-1,)).double().sum(),
                               eng.buffer("W4b", torch.bfloat16, (-1,)).double().abs().sum()])
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            replicas_consistent = bool(torch.equal(lo, hi))
            if rank == 0 and not replicas_consistent:
                print("bench.py: replicas diverged: %r vs %r" % (lo.tolist(), hi.tolist()), file=sys.stderr)
        # Not the headline: the same K steps again with the bf16 gradient payload (half the all-reduce
        # bytes), so that one run shows what the exchange costs at this GPU count.
        alt, alt_key, ddp_pick, alt_bf16, alt_sh = None, None, None, None, None
        if isinstance(runner, ddp.NativeDdpRunner) and runner.sharded and os.environ.get("RV_DDP_ALT", "1") == "1":
            alt_key = "alt_allreduce"
            # Not the headline: the same K steps with the all-reduce + full-update schedule on a second engine
            # (the sharded engine's moments are shard-local, so it cannot simply switch modes)
            ar_what = "fp32 all-reduce (2 buckets, fc4 | rest, behind backward), full Adam on every rank"
            try:
                eng2 = E.TrainEngine(S, H, L, B, device=dev, kl_beta=KL_BETA, lr=LR, seed=1000 + rank, ring=256)
                eng2.load_params(make_params(S, H, L, 0))
                run2 = ddp.NativeDdpRunner(eng2, comm, comp, sharded=False)
                for i in range(args.warmup + 1):
                    run2.step(pool[i % POOL])
                torch.cuda.synchronize()
                apasses = []
                for r in range(len(passes)):     # the same number of passes, bracketed the same way
                    if world > 1:
                        dist.barrier()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for i in range(args.steps):
                        run2.step(pool[(args.warmup + r * args.steps + i) % POOL])
                    torch.cuda.synchronize()
                    if world > 1:
                        dist.barrier()
                    torch.cuda.synchronize()
                    el = time.perf_counter() - t1
                    if world > 1:
                        t = torch.tensor([el], dtype=torch.float64, device=dev)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                        el = float(t.item())
                    apasses.append(el)
                apasses.sort()
                adt = apasses[len(apasses) // 2]
                a_consistent = None
                if world > 1:
                    chk = torch.stack([eng2.param.double().sum(), eng2.param.double().abs().sum()])
                    lo, hi = chk.clone(), chk.clone()
                    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                    a_consistent = bool(torch.equal(lo, hi))
                alt = {"grad_allreduce": ar_what, "ms_per_step": adt / args.steps * 1e3,
                       "value": float(B) * world * args.steps / adt, "repeats": len(apasses),
                       **({"replicas_consistent": a_consistent} if a_consistent is not None else {})}
                # side line only (it rounds each rank's summed gradient to bf16 before the exchange): the same
                # all-reduce schedule with half the bytes on the links
                try:
                    run2.set_payload("bf16")
                    for i in range(3):
                        run2.step(pool[i % POOL])
                    torch.cuda.synchronize()
                    bp = []
                    for r in range(min(5, len(passes))):
                        if world > 1:
                            dist.barrier()
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        for i in range(args.steps):
                            run2.step(pool[(r * args.steps + i) % POOL])
                        torch.cuda.synchronize()
                        if world > 1:
                            dist.barrier()
                        torch.cuda.synchronize()
                        el = time.perf_counter() - t1
                        if world > 1:
                            t = torch.tensor([el], dtype=torch.float64, device=dev)
                            dist.all_reduce(t, op=dist.ReduceOp.MAX)
                            el = float(t.item())
                        bp.append(el)
                    bp.sort()
                    alt_bf16 = {"grad_allreduce": "all-reduce schedule with the bf16 gradient payload (half the bytes; "
                                                  "gradients rounded to bf16 before the exchange)",
                                "ms_per_step": bp[len(bp) // 2] / args.steps * 1e3,
                                "value": float(B) * world * args.steps / bp[len(bp) // 2], "repeats": len(bp)}
                except Exception as exc:
                    alt_bf16 = {"grad_allreduce": "bf16 payload", "error": str(exc)[:200]}
                finally:
                    run2.set_payload("fp32")
                # Both schedules are the product's (same step, same arithmetic up to the order of the reduction);
                # which one is faster depends on the GPU count and the links, and this is the first hardware either
                # has run on at N > 1: the headline is the faster of the two, the other stays beside it.
                if adt < dt and a_consistent is not False and os.environ.get("RV_DDP_PICK", "0") == "1":
                    alt_key = "alt_sharded"
                    alt = {"grad_allreduce": ddp_mode, "ms_per_step": dt / args.steps * 1e3,
                           "value": float(B) * world * args.steps / dt, "repeats": len(passes),
                           **({"replicas_consistent": replicas_consistent} if replicas_consistent is not None else {})}
                    dt, passes, replicas_consistent = adt, apasses, a_consistent
                    dt_min, dt_max = apasses[0], apasses[-1]
                    ddp_mode = ar_what + ", issued by rv_plan_step_ddp on its own stream"
                    last = eng2.losses(min(8, args.steps))
                    ddp_pick = "all-reduce schedule (faster than the sharded optimizer in this run; both timed alike)"
                else:
                    ddp_pick = "sharded optimizer (not slower than the all-reduce schedule in this run; both timed alike)"
            except Exception as exc:   # the headline above is already measured: report, do not lose it
                alt = {"grad_allreduce": ar_what, "error": str(exc)[:200]}
        elif isinstance(runner, ddp.NativeDdpRunner) and os.environ.get("RV_DDP_ALT", "1") in ("1", "2"):
            # side line, never the headline: the same K steps with the sharded optimizer on a second engine.  On
            # several GPUs only with RV_DDP_ALT=2: a second schedule that has never run on more than one rank must
            # not be able to take the measured headline down with it (a failure inside a collective is a hang).
            try:
                if world > 1 and os.environ.get("RV_DDP_ALT", "1") != "2":
                    raise RuntimeError("skipped at N > 1 (set RV_DDP_ALT=2 to time it)")
                eng2 = E.TrainEngine(S, H, L, B, device=dev, kl_beta=KL_BETA, lr=LR, seed=1000 + rank, ring=256)
                eng2.load_params(make_params(S, H, L, 0))
                run2 = ddp.NativeDdpRunner(eng2, comm, comp, sharded=True)
                for i in range(args.warmup + 1):
                    run2.step(pool[i % POOL])
                torch.cuda.synchronize()
                sp_ = []
                for r in range(min(len(passes), 25)):
                    if world > 1:
                        dist.barrier()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for i in range(args.steps):
                        run2.step(pool[(args.warmup + r * args.steps + i) % POOL])
                    torch.cuda.synchronize()
                    if world > 1:
                        dist.barrier()
                    torch.cuda.synchronize()
                    el = time.perf_counter() - t1
                    if world > 1:
                        t = torch.tensor([el], dtype=torch.float64, device=dev)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                        el = float(t.item())
                    sp_.append(el)
                sp_.sort()
                s_consistent = None
                if world > 1:
                    ddp.gather_sharded_params(eng2)
                    chk = torch.stack([eng2.param.double().sum(), eng2.param.double().abs().sum()])
                    lo, hi = chk.clone(), chk.clone()
                    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                    s_consistent = bool(torch.equal(lo, hi))
                alt_sh = {"grad_allreduce": "sharded optimizer: fp32 reduce-scatter (fc4 | rest) -> Adam on this rank's 1/%d of the "
                                            "arena -> all-gather of the %s" % (world, "16-bit parameter message (bf16 weights + fp32 biases)"
                                                                               if getattr(eng2, "shard_gather", "fp32") == "bf16" else "fp32 parameters"),
                          "ms_per_step": sp_[len(sp_) // 2] / args.steps * 1e3,
                          "value": float(B) * world * args.steps / sp_[len(sp_) // 2], "repeats": len(sp_),
                          **({"replicas_consistent": s_consistent} if s_consistent is not None else {})}
                del run2, eng2
            except Exception as exc:
                alt_sh = {"grad_allreduce": "sharded optimizer", "error": str(exc)[:200]}
            try:
                runner.set_payload("bf16")
                for i in range(5):
                    one_step(i)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    one_step(i)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                adt = time.perf_counter() - t1
                if world > 1:
                    t = torch.tensor([adt], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    adt = float(t.item())
                alt = {"grad_allreduce": "bf16 payload, same schedule", "ms_per_step": adt / args.steps * 1e3,
                       "value": float(B) * world * args.steps / adt}
            except Exception as exc:   # the headline above is already measured: report, do not lose it
                alt = {"grad_allreduce": "bf16 payload, same schedule", "error": str(exc)[:200]}
            finally:
                runner.set_payload("fp32")
        if rank != 0:
            kern_ms, kern_flops, kern_cfg, kern_batches = None, None, None, None
        elif args.step_kernels_only:
            kern_ms, kern_flops, kern_cfg, kern_batches = time_dominant_kernel_batch(eng, pool[0])
        else:
            kern_ms, kern_flops, kern_cfg, kern_batches = time_dominant_kernel(eng, pool[0])
        # Side lines, never the headline: the same K steps on engines with opt-in reduced-precision storage
        alts = {}
        if world == 1 and not args.no_alts:
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
            alts["alt_fp8"] = time_alt(
                "fc1 and fc4 forward on e4m3 operands (v_mfma_scale_f32_16x16x128_f8f6f4, per-tensor scales, delayed "
                "activation scaling); backward and everything else bf16", fp8=True)
            try:
                alts["alt_deep_c4"] = time_deep_c4(dev, comp, max(10, args.steps // 4), 5)
            except Exception as exc:
                alts["alt_deep_c4"] = {"what": "deep variant (BASELINE configs[3])", "error": str(exc)[:200]}
            alts["alt_fp32_slabs"] = time_alt(
                "split-K partial sums of dW1 / dW4 stored as fp32 instead of block-floating-point fp16 (round 2's "
                "default); everything else as the headline", slab_dtype="fp32")

    if not all(map(lambda v: v == v and abs(v) < 1e3, last)):
        print("bench.py: non-finite loss %r" % (last,), file=sys.stderr)
        sys.exit(3)

    if rank == 0:
        frames = float(B) * world * args.steps
        value = frames / dt
        achieved = kern_flops / (kern_ms * 1e-3) / 1e12
        # HBM bytes per launch of the same kernel: NOT measured in this run -- the newest committed PMC summary
        # (profiles/rNN_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command,
        # corrected as MI355X_MICROARCH.md 'HBM' prescribes; tools/pmc_round.sh), labelled with its source
        traffic, traffic_src = None, None
        try:
            import glob
            cand = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]*_traffic.json")),
                          key=lambda f: os.path.basename(f).replace("_traffic", "_v0_traffic")
                          if "_v" not in os.path.basename(f) else os.path.basename(f))
            with open(cand[-1]) as f:
                traffic = json.load(f)["traffic_bytes"]
            traffic_src = "profiles/%s (PMC passes of an earlier builder run, not this run)" % os.path.basename(cand[-1])
        except Exception:
            pass
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
                       **({"ddp_mode": "sharded" if getattr(runner, "sharded", False) else "allreduce",
                           "shard_gather": getattr(eng, "shard_gather", None)} if runner is not None else {}),
                       "grad_allreduce": ddp_mode},
            # `value` / `ms_per_step` are the MEDIAN over `repeats` passes of exactly `steps` steps each
            "timing": {"repeats": len(passes), "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
                       "timed_seconds_total": sum(passes)},
            "host_us_per_step": host_dt / args.steps * 1e6,
            **({(alt_key or "alt_bf16_payload"): alt} if alt else {}),
            **({"alt_sharded": alt_sh} if alt_sh else {}),
            **({"ddp_schedule_pick": ddp_pick} if ddp_pick else {}),
            **({"alt_allreduce_bf16_payload": alt_bf16} if alt_bf16 else {}),
            **alts,
            **({"replicas_consistent": replicas_consistent} if replicas_consistent is not None else {}),
            **({"native_fallback_reason": native_fallback_reason, "rccl_version": getattr(comm, "version", None),
                "comm_stream_pick": [{"us_per_round_trip": u, "candidates_tried": n} for u, n in ddp.comm_stream_report()]}
               if world > 1 or force_ddp else {}),
            "step_tflops": value * flops_per_frame(S, H, L) / 1e12,
            "step_mfma_frac": value * flops_per_frame(S, H, L) / 1e12 / (PEAK_BF16_TFLOPS * world),
            # algorithmic HBM bytes per step: 54,784 B/frame of activations + 38 B/param
            "step_hbm_frac": (value * 54784.0 + (value / B / world) * 38.0 * 4592768 * world) / (8.0e12 * world),
            "final_loss": last[-1],
            "roofline": {"bound": "mfma", "kernel": kern_cfg,
                         "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "us_per_launch": kern_ms * 1e3,
                         "us_per_launch_batches": [b * 1e3 for b in kern_batches],
                         "timing": "in the step: median over 7 alternating replays of (hipGraph of 10 phase-by-phase steps WITH the "
                                   "plan's fc4-backward launch) - (the same graph WITHOUT it), / 10, HIP events after every "
                                   "replay; kernel boundaries included (back-to-back batches of this kernel alone slow the "
                                   "chip down: 33 -> 44 us over 250 launches)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle.torch_port import cpu_description, time_cpu_step
            fps, ms_c, n, threads = time_cpu_step(S, H, L, B, make_params(S, H, L, 0), make_frames(B, S, 1234),
                                                  seconds=args.cpu_seconds)
            # BASELINE configs[0] (the reference's own CPU-runnable case: 512-sample frames, latent 8, batch 32)
            s_fps, s_ms, s_n, _ = time_cpu_step(512, H, 8, 32, make_params(512, H, 8, 0), make_frames(32, 512, 1234),
                                                seconds=min(3.0, args.cpu_seconds), threads=threads)
            # the same C2 step on ALL the CPUs this process may use (SURVEY 8d asks for os.cpu_count() threads); a
            # one-GPU box can be a 16-CPU share of a 256-thread host, where this oversubscribes: bounded to a few steps
            all_cores = None
            try:
                usable = len(os.sched_getaffinity(0))
            except AttributeError:
                usable = os.cpu_count()
            if usable and usable != threads:
                try:
                    a_fps, a_ms, a_n, a_thr = time_cpu_step(S, H, L, B, make_params(S, H, L, 0), make_frames(B, S, 1234),
                                                          seconds=min(4.0, args.cpu_seconds), warmup=1, threads=usable, min_steps=1)
                    all_cores = {"value": a_fps, "unit": "frames/s", "cores": a_thr, "ms_per_step": a_ms,
                                 "sample": "%d steps of the same C2 step on every usable CPU" % a_n}
                except Exception as exc:
                    all_cores = {"error": str(exc)[:200]}
            out["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": threads, "kind": "port",
                                   **({"all_cores": all_cores} if all_cores else {}),
                                   "sample": "%d steps of the same C2 step (B=4096) in stock PyTorch fp32 on the "
                                             "host, median %.1f ms/step" % (n, ms_c),
                                   "smoke_shape": {"value": s_fps, "unit": "frames/s", "ms_per_step": s_ms,
                                                   "sample": "%d steps of S=512 H=2048 L=8 B=32" % s_n},
                                   **cpu_description()}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()   # rank 0 did extra timing work; leave together
        if comm is not None:
            torch.cuda.synchronize()
            comm.destroy()
        dist.destroy_process_group()
    if replicas_consistent is False:
        sys.exit(4)   # a data-parallel step that leaves the replicas with different weights is not a result


if __name__ == "__main__":
    main()
