"""Data-parallel training for the fused engine: one process per GPU, `torch.distributed` for the process group
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference has no distributed code; frames are independent, so data parallelism is a SUM of the
ranks' gradients per step with every rank on the same per-rank batch size: the mean of rank gradients is the gradient
of the global-batch mean loss (the reference's loss is a mean, rawvae/model.py:39,45), and Adam applies 1/world.

Two routes, same arithmetic:
  * `NativeDdpRunner` (the product path; train.py, bench.py): the LIBRARY issues the collectives --
    `rv_plan_step_ddp` gets the communicator handle and the address of ncclAllReduce, and runs the whole step from one
    host call on two streams (the caller's and a collective stream chosen by `pick_comm_stream`): buckets fc4 | the rest,
    each exchanged as soon as its gradients exist, fp32 payload by default (`DEFAULT_PAYLOAD`: the exact mean; "bf16" is
    opt-in and the bench's named choice), device-side flags between the two streams, fc4's Adam beside the second
    exchange.  DESIGN.md 5 has the schedule, its one-GPU model of 8 ranks and the one-rank RCCL rehearsal.  (The
    sharded-optimizer schedule of rounds 3-5 -- reduce-scatter, Adam on a shard, all-gather -- won no row of that model
    and was removed in round 6.)
  * `DdpRunner` + `GradSync`: the phases of `rv_plan_step` with `torch.distributed` all-reduces between them
    (works with any backend, gloo on CPU included): the route the CPU tests and the startup cross-check use.
"""
import ctypes as C
import os
import time

import torch
import torch.distributed as dist


class GradSync:
    """Bucketed asynchronous all-reduce over a flat gradient tensor.

    buckets: list of (lo, hi) element ranges of `flat`, in the order they become ready.
    """

    def __init__(self, flat, buckets, group=None):
        self.flat = flat
        self.buckets = [(int(lo), int(hi)) for lo, hi in buckets]
        for lo, hi in self.buckets:
            if not (0 <= lo < hi <= flat.numel()):
                raise ValueError("bucket (%d, %d) outside the gradient arena of %d elements"
                                 % (lo, hi, flat.numel()))
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a one-rank group still goes through the backend when asked to (RCCL rehearsal on a one-GPU box)
        self.active = dist.is_initialized() and (self.world > 1 or os.environ.get("RV_FORCE_DDP") == "1")
        self._pending = []

    def start(self, i):
        """Launch the all-reduce of bucket i (no-op without a process group)."""
        if not self.active:
            return
        lo, hi = self.buckets[i]
        self._pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM,
                                             group=self.group, async_op=True))

    def wait_one(self):
        """Wait for the oldest launched bucket only."""
        if self._pending:
            self._pending.pop(0).wait()

    def wait(self):
        """Make the current stream (GPU) or the host (CPU/gloo) wait for all launched buckets."""
        for w in self._pending:
            w.wait()
        self._pending = []

    @property
    def grad_scale(self):
        return 1.0 / self.world


# Gradient payload of the all-reduce schedule when the caller does not choose: "fp32" -- the exact mean of the ranks' fp32
# gradients, what a single process computes on the concatenated batch and what train.py / attach_comm give a
# user who asked for nothing.  "bf16" is OPT-IN (RV_DDP_PAYLOAD=bf16, attach_comm(payload="bf16")): each rank's fp32
# slab sum is rounded to bf16 once, RCCL sums in bf16, Adam reads the bf16 sum -- half the bytes of the one exchange
# nothing can hide, which is what the 1 -> 8 GPU scaling target is sized against (bench.py names
# it in its line and times it by explicit choice).  Its cost is known only from one GPU so far: 2^-9 relative per element
# and rank, the data-parallel identity to 5e-4 instead of 2e-5, a 20-step loss trajectory within 1e-4 of the fp32
# payload's (tests/test_ddp_gpu.py); no multi-GPU run has shown the gain or the convergence yet, hence not the default
# (round-4 advisor).
DEFAULT_PAYLOAD = "fp32"
# what bench.py exchanges at N > 1 unless RV_DDP_PAYLOAD says otherwise: named in its JSON line, fp32 timed beside it
BENCH_PAYLOAD = "bf16"

ARENA_SLACK = 1024   # elements every flat arena extends past n_params (16-byte accesses of the last tensor's tail)


def engine_buckets(engine):
    """Buckets in the order their gradients become available in backward:
    fc4 (after the paired fc4 backward), fc1 (end of the dependent chain), then fc21/fc22/fc3."""
    o = engine.offsets
    return [(o["fc4.weight"], engine.n_params), (0, o["fc21.weight"]), (o["fc21.weight"], o["fc4.weight"])]


def ddp_step(engine, sync, x, eps=None, stream=None):
    """One data-parallel training step (train.py:184-193 across ranks).

    Timeline on the compute stream (RCCL runs on its own stream, ordered by events):
        FWD, fc4 backward, finalize fc4            -> all-reduce fc4 (8.4 MB) starts
        dz, reparam bwd, heads dgrad, fc1 wgrad,
        finalize fc1                               -> all-reduce fc1 (8.4 MB) starts
        fc3 wgrad, heads wgrad, finalize rest      -> all-reduce rest (1.6 MB) starts
        wait fc4,  Adam(fc4)                       [overlaps the fc1 / rest all-reduces]
        wait fc1,  Adam(fc1)
        wait rest, Adam(fc21, fc22, fc3)
    so the fc4 exchange hides behind ~80 us of backward and the fc1 exchange behind the remaining
    weight-gradient GEMMs and the fc4 optimizer step.
    """
    from . import _lib as P
    engine.step(x, eps, phases=P.PHASE_FWD | P.PHASE_BWD_FC4 | P.PHASE_FIN_FC4, stream=stream)
    sync.start(0)
    engine.step(x, eps, phases=P.PHASE_BWD_CHAIN | P.PHASE_FIN_FC1, stream=stream)
    sync.start(1)
    engine.step(x, eps, phases=P.PHASE_BWD_REST | P.PHASE_FIN_MID, stream=stream)
    sync.start(2)
    g = sync.grad_scale
    sync.wait_one()
    engine.step(x, eps, phases=P.PHASE_ADAM_FC4, grad_scale=g, adam_from_flat=True, stream=stream)
    sync.wait_one()
    engine.step(x, eps, phases=P.PHASE_ADAM_FC1, grad_scale=g, adam_from_flat=True, stream=stream)
    sync.wait()
    engine.step(x, eps, phases=P.PHASE_ADAM_MID, grad_scale=g, adam_from_flat=True, stream=stream)


class DdpRunner:
    """`ddp_step` with its six compute segments replayed from hipGraphs (the all-reduces between
    them stay eager RCCL calls).  Only the first segment reads the batch, so it is captured once
    per distinct batch buffer; the other five are captured once."""

    def __init__(self, engine, sync, stream, use_graphs=True):
        from . import _lib as P
        self.engine, self.sync, self.stream, self.use_graphs = engine, sync, stream, use_graphs
        self._first = {}
        self._rest = None
        self._seg = [(P.PHASE_BWD_CHAIN | P.PHASE_FIN_FC1, False), (P.PHASE_BWD_REST | P.PHASE_FIN_MID, False),
                     (P.PHASE_ADAM_FC4, True), (P.PHASE_ADAM_FC1, True), (P.PHASE_ADAM_MID, True)]
        self._p_first = P.PHASE_FWD | P.PHASE_BWD_FC4 | P.PHASE_FIN_FC4

    def _capture(self, fn):
        from .engine import Graph
        g = Graph(self.stream)
        with g:
            fn()
        return g

    def step(self, x):
        e, s, st = self.engine, self.sync, self.stream
        if not self.use_graphs:
            return ddp_step(e, s, x, stream=st)
        scale = s.grad_scale
        key = x.data_ptr()
        if key not in self._first:
            self._first[key] = self._capture(lambda: e.step(x, phases=self._p_first, stream=st))
        if self._rest is None:
            self._rest = [self._capture(lambda ph=ph, fl=fl: e.step(x, phases=ph, grad_scale=scale,
                                                                    adam_from_flat=fl, stream=st))
                          for ph, fl in self._seg]
        self._first[key].launch()
        s.start(0)
        self._rest[0].launch()
        s.start(1)
        self._rest[1].launch()
        s.start(2)
        s.wait_one()
        self._rest[2].launch()
        s.wait_one()
        self._rest[3].launch()
        s.wait()
        self._rest[4].launch()
        e.host_steps += 1


def _loaded_rccl_path():
    """Path of the librccl this process has mapped already, or None."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    return line.split()[-1]
    except OSError:
        pass
    return None


class RcclComm:
    """An RCCL communicator owned by this package (one per process / GPU), created with ctypes from
    the RCCL library PyTorch already loaded, so `rv_plan_step_ddp` can issue the gradient all-reduces
    itself -- on its own stream, ordered by events, inside the same host call (and hipGraph) as the
    kernels.  The unique id travels over the existing `torch.distributed` group (any backend)."""

    class _UniqueId(C.Structure):
        _fields_ = [("internal", C.c_ubyte * 128)]

    def __init__(self, group=None, lib_path=None):
        if not dist.is_initialized():
            raise RuntimeError("RcclComm needs an initialised torch.distributed group to share the RCCL unique id")
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # Use the RCCL instance PyTorch itself has mapped (a second copy of the library would be a second,
        # unrelated collective runtime in this process): take its path from /proc/self/maps and open it with
        # RTLD_NOLOAD, which fails instead of loading anything new.  Only when torch has not loaded RCCL yet
        # (no "nccl" process group so far) is the library next to torch opened by path.
        loaded = _loaded_rccl_path()
        path = lib_path or os.environ.get("RV_RCCL_LIB") or loaded or \
            os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if loaded and os.path.realpath(path) == os.path.realpath(loaded):
            self._lib = C.CDLL(path, mode=os.RTLD_NOW | getattr(os, "RTLD_NOLOAD", 4))
        else:
            self._lib = C.CDLL(path)
            if loaded:
                raise RuntimeError("RcclComm: asked to open %s while the process already maps %s" % (path, loaded))
        self.lib_path = path
        v = C.c_int(0)
        try:
            self._lib.ncclGetVersion(C.byref(v))
        except AttributeError:
            pass
        self.version = int(v.value)
        self._lib.ncclGetErrorString.restype = C.c_char_p
        self._lib.ncclGetErrorString.argtypes = [C.c_int]
        uid = self._UniqueId()
        if self.rank == 0:
            self._check(self._lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        box = [bytes(bytearray(uid.internal)) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        C.memmove(C.byref(uid), box[0], 128)
        comm = C.c_void_p()
        self._lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, self._UniqueId, C.c_int]
        self._check(self._lib.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.handle = comm
        cnt = C.c_int(0)   # the communicator's own idea of its size (bench.py prints it: a one-rank rehearsal says 1)
        try:
            self._lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            self._check(self._lib.ncclCommCount(comm, C.byref(cnt)), "ncclCommCount")
        except AttributeError:
            cnt = C.c_int(self.world)
        self.rccl_count = int(cnt.value)
        self.allreduce_addr = C.cast(self._lib.ncclAllReduce, C.c_void_p)
        self._lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self._lib.ncclGetErrorString(rc).decode()))

    def all_reduce_(self, t, stream=None):
        """In-place SUM of a contiguous fp32 tensor on `stream` (self-test / utility)."""
        s = stream if stream is not None else torch.cuda.current_stream()
        self._check(self._lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), 7, 0, self.handle,
                                            s.cuda_stream or None), "ncclAllReduce")

    def self_test(self, device):
        """Every rank contributes rank+1 in each bucket-sized slot; the sum must be world(world+1)/2."""
        t = torch.full((1 << 16,), float(self.rank + 1), dtype=torch.float32, device=device)
        self.all_reduce_(t)
        torch.cuda.synchronize(device)
        want = self.world * (self.world + 1) / 2.0
        if not bool((t == want).all()):
            raise RuntimeError("RCCL self-test: expected %g everywhere, got [%g, %g]" % (want, t.min().item(), t.max().item()))

    def destroy(self):
        if getattr(self, "handle", None):
            self._lib.ncclCommDestroy.argtypes = [C.c_void_p]
            self._lib.ncclCommDestroy(self.handle)
            self.handle = None


_COMM_STREAMS = {}   # (device index, compute stream handle) -> (collective stream, us per ping-pong, candidates tried)


def _agree_max(value, device, group=None):
    """MAX of a host float over the ranks of `group` (identity without a process group): every rank gets the same
    answer, so a decision taken from it is taken by all ranks or by none."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    on_gpu = dist.get_backend(group) == "nccl"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def pick_comm_stream(compute_stream, device, tries=8, good_us=45.0, bad_us=100.0, group=None):
    """A high-priority stream for the collectives whose cross-stream waits against `compute_stream` stay on the
    device.  The HIP runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4); when two
    streams that wait on each other share one, the runtime resolves the waits on the host and every kernel behind
    them starts ~50 us late (measured: 880 instead of 255 us per data-parallel step at one rank, depending only on how
    many streams the process had created before).  So: time a short ping-pong (kernel, event, wait, kernel, event,
    wait) between the compute stream and a candidate; a healthy pair takes ~15 us per round trip, an affected one
    >100.  Candidates are created one after the other (each lands on the next hardware queue) until one is healthy;
    the rejected ones are kept alive so that the mapping of the chosen one does not move.  Cached per compute stream.

    If even the best candidate takes `bad_us` or more per round trip the call raises instead of silently running a
    step that is several times slower (RV_COMM_STREAM_ALLOW_SLOW=1 overrides).  With a process group the decision is
    COLLECTIVE: the ranks agree on the worst rank's best round trip (one MAX all-reduce over torch.distributed), so
    either every rank raises or none does -- a rank that raised alone would leave its peers inside the step's first
    RCCL collective with nobody to talk to.  Every rank of `group` must therefore call this at the same point (the
    engine does: NativeDdpRunner / the first step_ddp on a compute stream).  A rejected pick is not cached: a caller
    that catches the error and calls again measures again and raises again."""
    key = (device.index, compute_stream.cuda_stream)
    if key in _COMM_STREAMS:
        return _COMM_STREAMS[key][0]
    a = torch.zeros(4096, device=device)
    ev1, ev2 = torch.cuda.Event(), torch.cuda.Event()

    def round_trip_us(cand, n):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(n):
            with torch.cuda.stream(compute_stream):
                a.add_(1.0)
                ev1.record(compute_stream)
            cand.wait_event(ev1)
            with torch.cuda.stream(cand):
                a.add_(1.0)
                ev2.record(cand)
            compute_stream.wait_event(ev2)
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / n * 1e6
    tried, best = [], None
    for _ in range(tries):
        cand = torch.cuda.Stream(device=device, priority=-1)
        round_trip_us(cand, 5)
        us = round_trip_us(cand, 40)
        tried.append((cand, us))
        if best is None or us < best[1]:
            best = (cand, us)
        if us < good_us:
            break
    worst = _agree_max(best[1], device, group)       # the same number on every rank
    if worst >= bad_us and os.environ.get("RV_COMM_STREAM_ALLOW_SLOW") != "1":
        _REJECTED.append(tried)                      # keep the streams alive (the hardware-queue mapping must not move)
        from ._lib import RvError
        raise RvError("pick_comm_stream: no healthy collective stream against the compute stream on at least one rank "
                      "(this rank: best of %d candidates %.0f us per round trip; worst rank %.0f; a healthy pair takes "
                      "~15): cross-stream waits would be resolved on the host and every kernel of a data-parallel step "
                      "would start ~50 us late.  Raise GPU_MAX_HW_QUEUES, or set RV_COMM_STREAM_ALLOW_SLOW=1 to run "
                      "anyway." % (len(tried), best[1], worst))
    _COMM_STREAMS[key] = (best[0], best[1], tried)
    return best[0]


_REJECTED = []


def comm_stream_report():
    """[(us per ping-pong of the chosen stream, candidates tried)] for bench / train logs."""
    return [(round(v[1], 1), len(v[2])) for v in _COMM_STREAMS.values()]


class NativeDdpRunner:
    """The data-parallel step as ONE host call per batch (`rv_plan_step_ddp`, collectives included);
    with `use_graph` each distinct batch buffer's step is captured once into a hipGraph and replayed."""

    def __init__(self, engine, comm, stream, use_graph=False, payload=None, defer=False):
        """payload: "fp32" (default, `DEFAULT_PAYLOAD`: the exact mean) or "bf16" (opt-in).
        defer: every step leaves its last wait and update to the next one, whose cast launch goes out first
        (`TrainEngine.set_ddp_defer`); the loop calls `flush()` before it reads anything back."""
        self.engine, self.comm, self.stream, self.use_graph = engine, comm, stream, use_graph
        self.defer = bool(defer) and not use_graph
        if stream is not None:
            # before any collective of the step and before any capture: the choice times a few launches and is
            # agreed between the ranks (pick_comm_stream)
            engine._pick_comm_stream(stream)
        engine.attach_comm(comm, payload=payload)
        if self.defer:
            engine.set_ddp_defer(True)
        self._graphs = {}
        self.payload = engine.ddp_payload

    def set_payload(self, payload):
        """"fp32" or "bf16" gradient exchange (captured graphs are dropped: the payload is baked in)."""
        self.engine.set_ddp_payload(payload)
        self.payload = payload
        self._graphs = {}

    def flush(self):
        """Complete a deferred step (no-op otherwise)."""
        self.engine.ddp_flush(self.stream)

    def step(self, x):
        e = self.engine
        if not self.use_graph:
            return e.step_ddp(x, stream=self.stream)
        key = x.data_ptr()
        if key not in self._graphs:
            from .engine import Graph
            g = Graph(self.stream)
            with g:
                e.step_ddp(x, stream=self.stream)
            e.host_steps -= 1   # the capture did not execute
            self._graphs[key] = g
        self._graphs[key].launch()
        e.host_steps += 1
