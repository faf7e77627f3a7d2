#!/usr/bin/env python3
"""One-GPU model of the data-parallel step at N ranks (this pool has one-GPU boxes; the driver owns the 8-GPU runs).

The native step (`rv_plan_step_ddp`, all-reduce schedule) is given a stand-in all-reduce (tools/fake_collective.hip,
`fake_allreduce`) that exchanges nothing but holds `blocks` workgroups on the collective stream for

    latency + 2 (w - 1) / w * bytes / bus bandwidth

at exactly the points where RCCL would run -- so the fork / join structure, the kernels that run beside the collectives
and everything that is left exposed behind the last byte are the real ones, and only the links are modelled.  The
reference point: w = 8, 300 GB/s per GPU, 15 us per collective; target: modelled step
<= 8/6 x the local step (>= 6x weak scaling 1 -> 8 GPUs).

    hipcc --offload-arch=gfx950 -shared -fPIC tools/fake_collective.hip -o tools/libfakecoll.so   (build() does it)
    python tools/ddp_model.py            # the table committed as profiles/r04_ddp_model.txt
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rawaudiovae_kelsey_amd.engine import TrainEngine  # noqa: E402
from rawaudiovae_kelsey_amd.synth import make_frames, make_params  # noqa: E402

S, H, L, B = 1024, 2048, 64, 4096
HERE = os.path.dirname(os.path.abspath(__file__))


class FakeComm(C.Structure):
    _fields_ = [("blocks", C.c_int), ("threads", C.c_int), ("lds_bytes", C.c_int), ("world", C.c_int),
                ("latency_us", C.c_float), ("bus_gb_per_s", C.c_float)]


class Comm:
    """What engine.attach_comm needs of a communicator, backed by the timing stand-ins."""

    def __init__(self, lib, world, latency_us, bus, blocks=32, threads=256, lds=0):
        self.cfg = FakeComm(blocks, threads, lds, world, latency_us, bus)
        self.handle = C.cast(C.pointer(self.cfg), C.c_void_p)
        self.allreduce_addr = C.cast(lib.fake_allreduce, C.c_void_p)
        self.world, self.rank = world, 0


def time_steps(fn, st, n=300, warm=30, reps=3):
    with torch.cuda.stream(st):
        for i in range(warm):
            fn(i)
        st.synchronize()
        out = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for i in range(n):
                fn(i)
            st.synchronize()
            out.append((time.perf_counter() - t0) / n * 1e6)
    return sorted(out)[len(out) // 2]


def main():
    lib = C.CDLL(os.path.join(HERE, "libfakecoll.so"))
    xs = [torch.from_numpy(make_frames(B, S, 10 + i)).cuda() for i in range(8)]
    st = torch.cuda.Stream()

    def engine():
        e = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=1)
        e.load_params(make_params(S, H, L, 0))
        return e
    e0 = engine()
    local = time_steps(lambda i: e0.step(xs[i % 8], stream=st), st)
    n_fc4 = S * H + S
    n_rest = e0.n_params - n_fc4
    print("C2 step (S=1024 H=2048 L=64, per-GPU batch 4096), one MI355X, eager launches; median of 3 x 300 steps")
    print("local step (rv_plan_step):                                   %6.1f us" % local)
    print("target for >= 6x at 8 GPUs: modelled step <= 8/6 x local =    %6.1f us" % (local * 8 / 6))
    print("buckets: fc4 %d elements, rest %d elements (fc1, heads, fc3)" % (n_fc4, n_rest))
    print()
    print("%-7s %-9s %-10s %-8s | %-22s | %9s %8s %9s" % ("world", "bus GB/s", "latency us", "payload", "collectives: fc4 / rest (us)",
                                                           "us/step", "x local", "-> scaling"))
    del e0
    # (world, bus GB/s, latency us, payload); the first row is the reference point, the others its neighbourhood
    cfgs = [(8, 300.0, 15.0, "bf16"), (8, 300.0, 15.0, "fp32"),
            (8, 200.0, 15.0, "bf16"), (8, 400.0, 15.0, "bf16"), (8, 300.0, 25.0, "bf16"), (8, 300.0, 8.0, "bf16"),
            (8, 200.0, 25.0, "fp32"), (4, 300.0, 15.0, "bf16"), (2, 300.0, 15.0, "bf16"),
            (1, 0.0, 0.0, "bf16"), (1, 0.0, 0.0, "fp32")]
    if len(sys.argv) > 1:
        cfgs = [(int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3]), sys.argv[4])]
    # RV_MODEL_LDS: dynamic LDS of every stand-in workgroup in bytes (default 0).  With 65536 a CU that hosts one cannot
    # host a 256 x 256 GEMM block (128 KiB) beside it: the pessimistic reading of what a real collective's workgroups do
    lds = int(os.environ.get("RV_MODEL_LDS", "0"))
    nblk = int(os.environ.get("RV_MODEL_BLOCKS", "32"))
    if lds or nblk != 32:
        print("(stand-in workgroups: %d x 256 threads, %d bytes of LDS each)" % (nblk, lds))
    for world, bus, lat, payload in cfgs:
        e = engine()
        e.attach_comm(Comm(lib, world, lat, bus, blocks=nblk if bus > 0 else 0, lds=lds), payload=payload)
        e.set_ddp_defer(os.environ.get("RV_DDP_DEFER", "1") == "1")   # a step's last wait + update behind the next step's cast
        t = time_steps(lambda i: e.step_ddp(xs[i % 8], stream=st), st)
        e.ddp_flush()
        es = 2 if payload == "bf16" else 4
        if bus > 0:
            ar = [lat + 2.0 * (world - 1) / world * n * es / (bus * 1e3) for n in (n_fc4, n_rest)]
            coll = "%5.1f / %5.1f" % tuple(ar)
        else:
            coll = "(no stand-in: schedule only)"
        print("%-7d %-9s %-10s %-8s | %-28s | %9.1f %8.3f %8.2fx" % (world, "%.0f" % bus if bus else "-", "%.0f" % lat if bus else "-",
                                                                   payload, coll, t, t / local, world * local / t), flush=True)
        del e
    print()
    print("scaling = world x local / modelled step (weak scaling: per-GPU batch fixed).  The stand-in holds 32 workgroups of 256")
    print("threads for its duration on the collective stream; it moves no data, so HBM traffic of a real collective is not modelled.")


if __name__ == "__main__":
    main()
