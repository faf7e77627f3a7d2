#!/usr/bin/env python3
"""Soak of the library-driven data-parallel step with the deferred tail (RV_OPT_DDP_DEFER_TAIL) on a real one-rank RCCL
communicator: N steps on rotating batches, enqueued without a host sync, against the plain local step from the same
weights -- with the fp32 payload and the local split count the two are the same arithmetic, so parameters, both moments
and every recorded loss must be BIT-identical at the end, and no flag wait may have run out.

    python tools/soak_ddp.py [steps]      (default 20000)
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from rawaudiovae_kelsey_amd.synth import make_frames, make_params  # noqa: E402
from rawaudiovae_kelsey_amd import ddp  # noqa: E402
from rawaudiovae_kelsey_amd.engine import TrainEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    S, H, L, B = 1024, 2048, 64, 4096
    xs = [torch.from_numpy(make_frames(B, S, 40 + i)).cuda() for i in range(8)]
    st = torch.cuda.Stream()

    def fresh():
        e = TrainEngine(S, H, L, B, kl_beta=1e-4, lr=1e-4, seed=3, ring=64)
        e.load_params(make_params(S, H, L, 0))
        return e
    comm = ddp.RcclComm()
    comm.self_test(torch.device("cuda", 0))
    ref, de = fresh(), fresh()
    run = ddp.NativeDdpRunner(de, comm, st, payload="fp32", defer=True)
    de.set_ddp_w1_wide(False)
    t0 = time.time()
    worst = 0
    with torch.cuda.stream(st):
        for i in range(n):
            run.step(xs[i % 8])
            ref.step(xs[i % 8], stream=st)
            if i % 50 == 49:       # the loss ring holds 64 steps
                run.flush()
                a, b = de.drain_losses(), ref.drain_losses()
                worst += int(a != b)
            if i % 5000 == 4999:
                print("step %d  loss %.6f  %.1f s" % (i + 1, de.last_loss()[0], time.time() - t0), flush=True)
        run.flush()
    st.synchronize()
    same = all(torch.equal(getattr(de, k), getattr(ref, k)) for k in ("param", "exp_avg", "exp_avg_sq"))
    shadows = all(torch.equal(de.buffer(k, torch.bfloat16, (-1,)), ref.buffer(k, torch.bfloat16, (-1,))) for k in ("W1b", "Whb", "W3b", "W4b"))
    print("%d steps: parameters / moments identical: %s; shadows identical: %s; drains with a differing loss: %d; "
          "flag time-outs: %d; finite: %s" % (n, same, shadows, worst, de.ddp_timeouts(),
                                               bool(torch.isfinite(de.param).all())))
    comm.destroy()
    dist.destroy_process_group()
    if not (same and shadows and worst == 0 and de.ddp_timeouts() == 0):
        sys.exit(1)


if __name__ == "__main__":
    main()
