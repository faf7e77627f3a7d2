"""Stock-PyTorch fp32 CPU port of the reference training step (oracle; see
oracle/__init__.py).  This is what `bench.py` times as `cpu_baseline`
(kind "port") on the GPU node's host cores, and a second, independently
written checker for the numpy restatement.

It performs the same ATen operations in the same order as the reference
(`rawvae/model.py:19-47`, `train.py:163,184-193`): five `addmm`s, relu, exp,
tanh, `mse_loss`, autograd backward and `torch.optim.Adam`; the one deliberate
difference is that eps is an argument instead of `torch.randn_like`.
"""
import time

import torch
import torch.nn.functional as F

from .inputs import PARAM_NAMES


class PortVAE(torch.nn.Module):
    def __init__(self, S, H, L):
        super().__init__()
        self.S, self.H, self.L = S, H, L
        self.fc1 = torch.nn.Linear(S, H)
        self.fc21 = torch.nn.Linear(H, L)
        self.fc22 = torch.nn.Linear(H, L)
        self.fc3 = torch.nn.Linear(L, H)
        self.fc4 = torch.nn.Linear(H, S)

    def load_numpy(self, params):
        with torch.no_grad():
            sd = self.state_dict()
            for k in PARAM_NAMES:
                sd[k].copy_(torch.from_numpy(params[k]))
        return self

    def forward(self, x, eps=None):
        x = x.view(-1, self.S)
        h1 = F.relu(self.fc1(x))
        mu, logvar = self.fc21(h1), self.fc22(h1)
        std = torch.exp(0.5 * logvar)
        if eps is None:
            eps = torch.randn_like(std)
        z = mu + eps * std
        recon = torch.tanh(self.fc4(F.relu(self.fc3(z))))
        return recon, mu, logvar


def port_loss(recon, x, mu, logvar, kl_beta, S):
    mse = F.mse_loss(recon, x.view(-1, S))
    kld = -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())
    return mse + kl_beta * kld


def cpu_description():
    """What the CPU baseline ran on: model string, logical CPUs the OS reports, CPUs this process may use,
    torch version."""
    import os
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "cpus_usable": usable,
            "torch_version": torch.__version__,
            "threads_note": "`cores` = torch threads used: min(usable CPUs, RV_CPU_THREADS or 16) -- a one-GPU box "
                            "is a 16-CPU share of a larger host"}


def time_cpu_step(S, H, L, B, params, x, seconds=15.0, warmup=2, kl_beta=1e-4, lr=1e-4,
                  threads=None, min_steps=3):
    """Time zero_grad/forward/loss/backward/Adam.step on the host CPU.
    Returns (frames_per_s, ms_per_step_median, steps_timed, threads)."""
    import os
    if not threads:
        try:
            threads = len(os.sched_getaffinity(0))
        except AttributeError:
            threads = os.cpu_count()
        # a one-GPU box is given a 16-core share of a much larger host; more threads than
        # that only oversubscribes (measured: 256 threads -> 12.7 s/step)
        threads = min(threads, int(os.environ.get("RV_CPU_THREADS", "16")))
    torch.set_num_threads(threads)
    model = PortVAE(S, H, L).load_numpy(params)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    xt = torch.from_numpy(x)
    times = []
    t_end = None
    i = 0
    while True:
        t0 = time.perf_counter()
        opt.zero_grad()
        recon, mu, logvar = model(xt)
        loss = port_loss(recon, xt, mu, logvar, kl_beta, S)
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        i += 1
        if i <= warmup:
            if i == warmup:
                t_end = time.perf_counter() + seconds
            continue
        times.append(dt)
        if time.perf_counter() >= t_end and len(times) >= min_steps:
            break
    times.sort()
    med = times[len(times) // 2]
    return B / med, med * 1e3, len(times), threads
