"""Data-parallel gradient exchange for the fused engine: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference has no distributed code; frames are independent, so the
exchange is one SUM all-reduce of the flat fp32 gradient arena per step, issued in two
buckets so that the first (fc3, fc4 -- ready after the first half of backward) overlaps the
rest of backward.  Every rank holds the same per-rank batch size, so mean-of-rank-gradients
equals the gradient of the global-batch mean loss (the reference's loss is a mean,
rawvae/model.py:39,45); Adam then applies `grad_scale = 1/world`.
"""
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
        self._pending = []

    def start(self, i):
        """Launch the all-reduce of bucket i (no-op for a single rank)."""
        if self.world == 1:
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


def engine_buckets(engine):
    """[fc3, fc4] first (ready after PHASE_BWD_A), then [fc1, fc21, fc22]."""
    cut = engine.offsets["fc3.weight"]
    return [(cut, engine.n_params), (0, cut)]


def ddp_step(engine, sync, x, eps=None, stream=None):
    """One data-parallel training step (train.py:184-193 across ranks).

    Timeline on the compute stream (RCCL runs on its own stream, ordered by events):
        FWD, BWD_A, finalize A        -> all-reduce A (fc3, fc4) starts
        BWD_B, finalize B             -> all-reduce B (fc1, heads) starts   [A overlaps this compute]
        wait A, Adam(fc3, fc4)                                             [overlaps all-reduce B]
        wait B, Adam(fc1, heads)
    """
    from ._lib import (PHASE_ADAM_A, PHASE_ADAM_B, PHASE_BWD_A, PHASE_BWD_B, PHASE_FINALIZE_A,
                       PHASE_FINALIZE_B, PHASE_FWD)
    engine.step(x, eps, phases=PHASE_FWD | PHASE_BWD_A | PHASE_FINALIZE_A, stream=stream)
    sync.start(0)
    engine.step(x, eps, phases=PHASE_BWD_B | PHASE_FINALIZE_B, stream=stream)
    sync.start(1)
    sync.wait_one()
    engine.step(x, eps, phases=PHASE_ADAM_A, grad_scale=sync.grad_scale, adam_from_flat=True, stream=stream)
    sync.wait()
    engine.step(x, eps, phases=PHASE_ADAM_B, grad_scale=sync.grad_scale, adam_from_flat=True, stream=stream)
