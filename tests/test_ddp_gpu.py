"""Two-rank rehearsal of the data-parallel bench flow on ONE GPU (gloo carries the all-reduces
through the host; on a multi-GPU node the same code runs with backend "nccl" = RCCL): phases,
three gradient buckets, per-bucket Adam, replica consistency."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import REPO  # noqa: E402


def _free_port():
    """A TCP port nobody listens on right now (the rendezvous of a torch.distributed.run child)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return str(so.getsockname()[1])


def test_two_rank_bench_flow_keeps_replicas_identical():
    env = dict(os.environ, RV_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(REPO, "bench.py"),
           "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "diverged" not in r.stderr, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8192 and out["scaling"] == "weak"
    assert out["value"] > 0 and 0 < out["final_loss"] < 1
    assert "cpu_baseline" not in out          # reported at N=1 only


def test_two_rank_bench_flow_with_the_library_driven_step_through_standins():
    """The N > 1 branch of bench.py AS THE DRIVER WILL RUN IT -- communicator, startup check against the torch.distributed
    route, the library-driven all-reduce step with its deferred tail, the flush that ends a timed pass, the fp32-payload
    side line on the same engine, the replica check, the JSON line -- with two ranks on this one GPU: RCCL refuses that,
    so the collectives are the functional stand-ins (RV_DDP_REHEARSAL=shm).  A rehearsal of the flow, not a measurement."""
    env = dict(os.environ, RV_DIST_BACKEND="gloo", RV_DDP_REHEARSAL="shm")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(REPO, "bench.py"),
           "--gpus", "2", "--steps", "6", "--warmup", "2", "--repeats", "3", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8192 and out["scaling"] == "weak"
    assert out["native_fallback_reason"] is None, out["native_fallback_reason"]
    assert "replicas identical" in out["startup_check"], out["startup_check"]
    assert out["config"]["ddp_mode"] == "allreduce" and out["config"]["ddp_payload"] == "bf16"
    assert "behind the next step's cast" in out["config"]["grad_allreduce"]      # the deferred tail is what was timed
    assert out["replicas_consistent"] is True and "rehearsal" in out
    assert out["alt_fp32_payload"]["ms_per_step"] > 0
    assert out["value"] > 0 and 0 < out["final_loss"] < 1


@pytest.mark.parametrize("world", [2, 4])
def test_native_ddp_step_two_processes_one_gpu(world):
    """`rv_plan_step_ddp` with world = 2 and 4 (every rank a process of its own on this one GPU): RCCL refuses two ranks on a
    device, so the collectives are the functional stand-ins of tools/fake_collective.hip (`shm_*`, RCCL's signatures, a
    real exchange through shared memory) -- everything else is the product path.  Both payloads of the all-reduce schedule
    (fp32 / bf16), both forms of fc1's weight gradient, with and without the deferred tail, small shape and C2:
    replicas identical, and equal to the torch.distributed route (tests/ddp_shm_worker.py) -- bit for bit with two
    ranks, to fp32 summation order with four (the two routes add the ranks' gradients in different orders); with and
    without the deferred tail; and the native step against the ORACLE on the concatenated batch of all ranks."""
    so = os.path.join(REPO, "tools", "libfakecoll.so")
    assert os.path.exists(so), "tools/libfakecoll.so missing: __graft_entry__.build() compiles it"
    env = dict(os.environ, RV_COMM_STREAM_ALLOW_SLOW="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(REPO, "tests", "ddp_shm_worker.py")]
    # (Two processes time-share ONE GPU here, which exposed an ordering bug the one-process tests never showed: an engine
    # initialised on one stream and stepped on another without an edge between the two -- engine._note_init /
    # _await_init.  About one run in fifteen failed before that fix; 45 consecutive runs passed after it.)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    if not (r.returncode == 0 and "DDP_SHM_OK" in r.stdout):
        keep = [l for l in (r.stdout + "\n" + r.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l and "hostname of the client" not in l]
        print("\n".join(keep[-80:]))
        try:   # (gpurun merges gpurun_out/ back: the full output survives the box)
            os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
            with open(os.path.join(REPO, "gpurun_out", "ddp_shm_fail.log"), "w") as f:
                f.write(r.stdout + "\n==== stderr ====\n" + r.stderr)
        except OSError:
            pass
    assert r.returncode == 0 and "DDP_SHM_OK" in r.stdout
    # HIP against the oracle (the worker's last section): the two-process native step on per-rank batches equals
    # oracle.train_step on the concatenated batch, at both shapes
    assert r.stdout.count("DDP_VS_ORACLE_OK") == 2, r.stdout[-2000:]
    # ... and with the bf16 gradient payload (bench.py's choice at N > 1) under both accumulation orders -- fp32 with one
    # rounding, and bf16 hop by hop in ring order as a ring all-reduce in the payload's type sums it (world - 1 roundings)
    # -- 7e-3 rel-L2 per tensor against the same oracle; the measured worst tensor is printed per order and shape
    lines = [l for l in r.stdout.splitlines() if l.startswith("DDP_BF16_PAYLOAD_VS_ORACLE_OK")]
    assert len(lines) == 4 and sum("ring-order" in l for l in lines) == 2, r.stdout[-3000:]
    print("\n".join(lines))


def test_native_rccl_step_one_rank_equals_local_step():
    """`rv_plan_step_ddp` (the library issues the RCCL all-reduces itself) with a real one-rank RCCL
    communicator: eager and as a replayed hipGraph it must produce exactly the parameters of the plain
    fused step (same kernels; sum over one rank and the 1/world mean are identities)."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", %r
dist.init_process_group("gloo", rank=0, world_size=1)
from oracle.inputs import make_frames, make_params
from rawaudiovae_kelsey_amd import ddp
from rawaudiovae_kelsey_amd.engine import Graph, TrainEngine
S, H, L, B = 512, 1024, 16, 256
def fresh():
    e = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=3)
    e.load_params(make_params(S, H, L, 0))
    return e
x = torch.from_numpy(make_frames(B, S, 1)).cuda()
st = torch.cuda.Stream()
ref = fresh()
with torch.cuda.stream(st):
    for _ in range(4):
        ref.step(x, stream=st)
st.synchronize()
comm = ddp.RcclComm()
comm.self_test(torch.device("cuda", 0))
eager = fresh(); eager.attach_comm(comm, payload="fp32"); eager.set_ddp_w1_wide(False)   # the local step's arithmetic
assert ddp.DEFAULT_PAYLOAD == "fp32" and TrainEngine.ddp_payload_default() == "fp32" and comm.rccl_count == 1
with torch.cuda.stream(st):
    for _ in range(4):
        eager.step_ddp(x, stream=st)
st.synchronize()
assert torch.equal(eager.param, ref.param), "eager native step differs"
assert eager.losses(4) == ref.losses(4)
# 300 more steps enqueued without a host sync (events and the collective stream are reused while
# earlier steps are still in flight): still bit-identical to the local step
with torch.cuda.stream(st):
    for _ in range(300):
        eager.step_ddp(x, stream=st)
        ref.step(x, stream=st)
st.synchronize()
assert torch.equal(eager.param, ref.param) and torch.equal(eager.exp_avg_sq, ref.exp_avg_sq)
ref = fresh()
with torch.cuda.stream(st):
    for _ in range(4):
        ref.step(x, stream=st)
st.synchronize()
gr = fresh()
run = ddp.NativeDdpRunner(gr, comm, st, use_graph=True, payload="fp32"); gr.set_ddp_w1_wide(False)
with torch.cuda.stream(st):
    for _ in range(4):
        run.step(x)
st.synchronize()
assert torch.equal(gr.param, ref.param), "graph-replayed native step differs"
assert gr.steps_done() == 4
# the fp8 weight path at C2 (all four large GEMM launches on e4m3 operands, fc1's weight gradient with its finalize riders
# included): the same bits as the local fp8 step, delayed scales and all
S8, H8, L8, B8 = 1024, 2048, 64, 4096
def fresh8():
    e = TrainEngine(S8, H8, L8, B8, kl_beta=1e-4, lr=1e-4, seed=3, fp8=True)
    e.load_params(make_params(S8, H8, L8, 0))
    return e
x8 = torch.from_numpy(make_frames(B8, S8, 1)).cuda()
r8, d8 = fresh8(), fresh8()
d8.attach_comm(comm, payload="fp32"); d8.set_ddp_w1_wide(False)
with torch.cuda.stream(st):
    for _ in range(4):
        r8.step(x8, stream=st)
        d8.step_ddp(x8, stream=st)
st.synchronize()
assert torch.equal(d8.param, r8.param) and d8.losses(4) == r8.losses(4), "fp8 native step differs"
assert d8.fp8_state()[13] == r8.fp8_state()[13] != 56.0 * B8 * S8      # dP1's scale: latched from a measurement, in both
assert not d8.buffer("dP1", torch.bfloat16, (-1,)).any()                  # and its bf16 copy was never written
del r8, d8
# bf16 payload: the summed gradient is rounded to bf16 before the exchange; Adam's first steps move every
# weight by ~lr whatever the gradient's magnitude, so the parameters stay within a fraction of lr of the
# fp32-payload run (sign flips of near-zero gradients aside) and the loss trajectory within 1e-4
df = fresh(); df.attach_comm(comm)      # the default: the exact fp32 mean
assert df.ddp_payload == "fp32"
del df
bf = fresh(); bf.attach_comm(comm, payload="bf16")      # opt-in
assert bf.ddp_payload == "bf16"
with torch.cuda.stream(st):
    for _ in range(4):
        bf.step_ddp(x, stream=st)
st.synchronize()
d = (bf.param - ref.param).abs()
assert float(d.max()) <= 2.1 * 4 * 1e-4 and float(d.mean()) < 0.02 * 4 * 1e-4, (float(d.max()), float(d.mean()))
assert not torch.equal(bf.param, ref.param)
for a, b in zip(bf.losses(4), ref.losses(4)):
    assert abs(a - b) <= 1e-4 * abs(b)
bf.set_ddp_payload("fp32")
# deferred tail (RV_OPT_DDP_DEFER_TAIL): each step leaves its last wait + update to the next call, whose cast launch goes
# out first; different batches per step (the early cast must not disturb the previous step's weight gradient), both
# payloads; flushed by the runner, by the health check, by a local step and by the state-dict readers -- always the
# bits of the undeferred run
xs = [torch.from_numpy(make_frames(B, S, 20 + i)).cuda() for i in range(6)]
for payload in ("fp32", "bf16"):
    plain = fresh(); plain.attach_comm(comm, payload=payload)
    de = fresh(); run = ddp.NativeDdpRunner(de, comm, st, payload=payload, defer=True)
    assert run.defer
    with torch.cuda.stream(st):
        for xi in xs:
            plain.step_ddp(xi, stream=st)
            run.step(xi)
    st.synchronize()
    assert not torch.equal(de.param, plain.param)          # the last update of fc1 / heads / fc3 is still pending
    # ... fc4's (the arena's tail) is not
    assert torch.equal(de.param[-(S * H + S):], plain.param[-(S * H + S):])
    with torch.cuda.stream(st):
        run.flush()
        run.flush()                                         # a second flush is a no-op
    st.synchronize()
    for name in ("param", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(de, name), getattr(plain, name)), (payload, name)
    assert de.losses(6) == plain.losses(6) and de.steps_done() == 6
    for name in ("W1b", "Whb", "W3b", "W4b"):
        assert torch.equal(de.buffer(name, torch.bfloat16, (-1,)), plain.buffer(name, torch.bfloat16, (-1,))), name
    with torch.cuda.stream(st):
        run.step(xs[0]); plain.step_ddp(xs[0], stream=st)
        assert de.ddp_timeouts() == 0                       # flushes first (the deferred wait counts)
        run.step(xs[1]); plain.step_ddp(xs[1], stream=st)
        de.step(xs[2], stream=st); plain.step(xs[2], stream=st)     # a local step completes the deferred one itself
        run.step(xs[3]); plain.step_ddp(xs[3], stream=st)
        sd = de.optimizer_state_dict()                      # so do the state-dict readers
    st.synchronize()
    assert torch.equal(de.param, plain.param) and torch.equal(de.exp_avg_sq, plain.exp_avg_sq), payload
    de.set_ddp_defer(False)
    with torch.cuda.stream(st):
        run.step(xs[4]); plain.step_ddp(xs[4], stream=st)
    st.synchronize()
    assert torch.equal(de.param, plain.param), payload
    del plain, de, run
comm.destroy()
dist.destroy_process_group()
print("NATIVE_OK")
''' % (REPO, _free_port())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0 and "NATIVE_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_flag_wait_behind_a_slow_collective_times_out_loudly_only_when_the_bound_is_short():
    """The compute stream's waits for the two exchanges depend on the slowest PEER, so their bound is minutes by default
    (RV_OPT_DDP_WAIT_MS); a wait that does run out must surface as an error, not as a silently wrong step.  The
    collective here is the timing stand-in of tools/fake_collective.hip with 30 ms of latency per call."""
    import ctypes as C
    import torch
    from rawaudiovae_kelsey_amd import _lib
    from rawaudiovae_kelsey_amd._lib import lib
    from rawaudiovae_kelsey_amd.engine import TrainEngine
    from rawaudiovae_kelsey_amd.synth import make_frames, make_params
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import ddp_model
    fake = C.CDLL(os.path.join(REPO, "tools", "libfakecoll.so"))
    S, H, L, B = 256, 512, 16, 128
    x = torch.from_numpy(make_frames(B, S, 3)).cuda()
    st = torch.cuda.Stream()

    def run(wait_ms):
        e = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=1)
        e.load_params(make_params(S, H, L, 0))
        e.attach_comm(ddp_model.Comm(fake, 2, 30000.0, 300.0, blocks=4), payload="fp32")
        lib().rv_plan_set_option(e._plan, _lib.OPT_DDP_SIGNAL, 1)     # device-side flags, whatever RV_DDP_SIGNAL says
        if wait_ms:
            lib().rv_plan_set_option(e._plan, _lib.OPT_DDP_WAIT_MS, wait_ms)
        with torch.cuda.stream(st):
            for _ in range(2):
                e.step_ddp(x, stream=st)
        st.synchronize()
        return e
    e = run(0)                      # default bound (30 s): the 30 ms collectives are simply waited for
    assert e.steps_done() == 2 and e.ddp_timeouts() == 0
    start = torch.from_numpy(__import__("numpy").concatenate([v.ravel() for v in make_params(S, H, L, 0).values()])).cuda()
    assert not torch.equal(e.param, start)          # (it trained)
    e = run(5)                      # 5 ms: every wait behind a collective runs out
    assert e.ddp_timeouts() > 0
    with pytest.raises(_lib.RvError, match="timed out"):
        e.steps_done()
    # ... and the plan is poisoned on the device: the updates behind the waits that ran out were NOT applied (a partial
    # all-reduce never reaches the weights), nor the moments touched
    assert torch.equal(e.param, start)
    assert not e.exp_avg.any() and not e.exp_avg_sq.any()
