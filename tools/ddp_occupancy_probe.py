#!/usr/bin/env python3
"""How sensitive is the data-parallel step to a collective kernel that holds CUs beside it?  One GPU: the
native step (rv_plan_step_ddp) is given a stand-in all-reduce (tools/fake_collective.hip) that occupies
`blocks` workgroups for `us` microseconds on the collective stream at the points where RCCL would run.
    hipcc --offload-arch=gfx950 -shared -fPIC tools/fake_collective.hip -o tools/libfakecoll.so
    python tools/ddp_occupancy_probe.py
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rawaudiovae_kelsey_amd.synth import make_frames, make_params  # noqa: E402
from rawaudiovae_kelsey_amd.engine import TrainEngine  # noqa: E402

S, H, L, B = 1024, 2048, 64, 4096


class FakeComm(C.Structure):
    _fields_ = [("blocks", C.c_int), ("threads", C.c_int), ("latency_us", C.c_int), ("kb_per_us", C.c_int),
                ("lds_bytes", C.c_int)]


class Comm:
    def __init__(self, lib, blocks, threads, latency_us, kb_per_us, lds=0):
        self.cfg = FakeComm(blocks, threads, latency_us, kb_per_us, lds)
        self.handle = C.cast(C.pointer(self.cfg), C.c_void_p)
        self.allreduce_addr = C.cast(lib.fake_allreduce, C.c_void_p)
        self.world = 1


def main():
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfakecoll.so"))
    xs = [torch.from_numpy(make_frames(B, S, 10 + i)).cuda() for i in range(4)]
    st = torch.cuda.Stream()
    # (blocks, threads, latency us, KiB per us, payload): an all-reduce of n bytes takes latency + n / rate;
    # 20 us + 150 KiB/us puts an 8.4 MB fp32 bucket at ~75 us, 300 KiB/us at ~48 us
    cfgs = ((0, 256, 0, 0, "fp32"), (32, 256, 20, 300, "fp32"), (32, 256, 20, 150, "fp32"), (32, 256, 20, 150, "bf16"),
            (32, 256, 20, 75, "fp32"), (32, 256, 20, 75, "bf16"))
    if len(sys.argv) > 1:
        cfgs = (tuple(int(v) for v in sys.argv[1:5]) + (sys.argv[5],),)
    for blocks, threads, lat, rate, payload in cfgs:
        e = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=1)
        e.load_params(make_params(S, H, L, 0))
        e.attach_comm(Comm(lib, blocks, threads, lat, rate))
        e.set_ddp_payload(payload)
        with torch.cuda.stream(st):
            for i in range(30):
                e.step_ddp(xs[i % 4], stream=st)
            st.synchronize()
            t0 = time.perf_counter()
            for i in range(300):
                e.step_ddp(xs[i % 4], stream=st)
            st.synchronize()
            dt = (time.perf_counter() - t0) / 300 * 1e6
        print("stand-in collective: %3d blocks, %2d us + bytes / %3d KiB/us, %s payload  ->  %6.1f us/step"
              % (blocks, lat, rate, payload, dt), flush=True)
        del e


if __name__ == "__main__":
    main()
