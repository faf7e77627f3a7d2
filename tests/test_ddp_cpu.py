"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: bucketed all-reduce of the
flat gradient arena, and the identity that makes DDP exact for this loss (mean over ranks of
per-rank gradients == gradient of the global-batch mean loss)."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

from conftest import REPO  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import vae_oracle as O
    from oracle.inputs import PARAM_NAMES, make_eps, make_frames, make_params
    from rawaudiovae_kelsey_amd.ddp import GradSync
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, H, L, B = 32, 48, 4, 8
        p = O.cast_params(make_params(S, H, L, 0), np.float64)
        x = make_frames(B * world, S, 3).astype(np.float64)
        eps = make_eps(B * world, L, 4).astype(np.float64)
        sl = slice(rank * B, (rank + 1) * B)
        c = O.forward(p, x[sl], eps[sl])
        g = O.backward(p, c, 1e-4)
        flat = torch.from_numpy(np.concatenate([g[k].reshape(-1) for k in PARAM_NAMES]))
        sizes = [g[k].size for k in PARAM_NAMES]
        c1, c4 = sum(sizes[:2]), sum(sizes[:8])   # engine_buckets(): fc4 | fc1 | the rest
        sync = GradSync(flat, [(c4, flat.numel()), (0, c1), (c1, c4)])
        sync.start(0)
        sync.start(1)
        sync.start(2)
        sync.wait_one()
        sync.wait()
        mean = flat.numpy() * sync.grad_scale
        cf = O.forward(p, x, eps)
        gf = O.backward(p, cf, 1e-4)
        full = np.concatenate([gf[k].reshape(-1) for k in PARAM_NAMES])
        err = float(np.abs(mean - full).max() / np.abs(full).max())
        q.put((rank, err, sync.world))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_equals_global_batch_gradient():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, w in res:
        assert w == 2 and err < 1e-12, (rank, err)


def test_gradsync_rejects_bad_buckets_and_is_noop_single_rank():
    from rawaudiovae_kelsey_amd.ddp import GradSync
    flat = torch.arange(10, dtype=torch.float32)
    with pytest.raises(ValueError):
        GradSync(flat, [(0, 11)])
    s = GradSync(flat, [(5, 10), (0, 5)])
    s.start(0)
    s.start(1)
    s.wait()
    assert s.grad_scale == 1.0 and torch.equal(flat, torch.arange(10, dtype=torch.float32))
