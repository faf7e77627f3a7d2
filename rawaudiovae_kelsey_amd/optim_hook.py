"""`torch.optim.Adam.step()` on a VAE's parameters as ONE fused launch.

The reference's training loop ends with `optimizer.step()` on `optim.Adam(model.parameters(), lr=...)`
(train.py:163,193).  Stock `torch.optim.Adam` runs its default "foreach" implementation: ~12 multi-tensor kernels over
the ten parameters, their two moment tensors and their gradients (~100 us of GPU time at C2) enqueued by ~235 us of
Python -- the largest single item of the drop-in loop's host time (profiles/r04_api_prof.txt).  The fused engine
already has the same update as one kernel that also refreshes the bf16 operand shadows (`rv_adam_multi`).

This module lets the UNCHANGED loop use it: a global optimizer step pre-hook (torch.optim.optimizer.
register_optimizer_step_pre_hook) recognises a param group of a plain `torch.optim.Adam` that holds all ten parameters
of a `rawvae.model.VAE` whose forward ran through the one-node path (fused.py: the Parameters are views of one fp32
arena), with hyper-parameters `rv_adam_multi` implements (betas (0.9, 0.999), eps 1e-8, no weight decay / amsgrad /
maximize / capturable / differentiable -- the reference's construction), performs the update with ONE launch, and hides
those parameters' `.grad` for the duration of the stock step (which skips parameters without a gradient, as it always
does), putting the gradients back in the post-hook.  The group's parameter list is never touched: if the stock step
raises, `zero_grad()` / `state_dict()` / `add_param_group()` see the optimizer whole (round-4 advisor; round 4 swapped
the list and restored it in the post-hook, which a raising step never reaches).  Everything else -- other optimizers,
other parameter groups, other parameters of the same group, other hyper-parameters, closures -- is left to PyTorch
untouched.

torch entry points this module leans on beyond the documented optimizer-hook API are feature-tested at import
(`_probe`): `torch.autograd.graph.increment_version` (public since 2.2; without it the hook is OFF and every step is
PyTorch's own -- the public path) and `torch._foreach_add_` (private name; without it the ten step counters are bumped
one by one with `Tensor.add_`).  `missing` says what was not found.

What stays PyTorch's: `optimizer.state[p]` keeps torch's layout (`step`, `exp_avg`, `exp_avg_sq`), so
`optimizer.state_dict()` / `load_state_dict()` and the reference's checkpoints (train.py:208-212) are unchanged; the
moment tensors ARE views of the engine's moment arenas (state loaded from a checkpoint is copied into them on the
next step).  The update differs from torch's by the hardware sqrt / reciprocal (~3e-7 relative; tests/test_model_gpu.py).
`rawaudiovae_kelsey_amd.optim_hook.enabled = False` (or RV_OPTIM_HOOK=0) switches it off.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib
from ._lib import lib, ptr, stream_ptr

enabled = os.environ.get("RV_OPTIM_HOOK", "1") != "0"
stats = {"fused_steps": 0, "declined": {}}   # how often the hook took a step over / why it left one to PyTorch
_installed = False
_OWNER = {}            # id(Parameter) -> (weakref(Parameter), weakref(module))
_STASH = weakref.WeakKeyDictionary()   # optimizer -> [(parameters, their gradients), ...] while the stock step runs
_T_RING = {}           # device -> int64 tensor [1, 2, 3, ...]: the step number is passed as a pointer into it


def _probe(t=torch):
    """Which of the torch entry points beyond the documented hook API exist in this torch: -> (bump_versions or None,
    add_one_to_each, [names not found]).  `bump_versions(params)` marks tensors as modified in place for autograd;
    without it the hook cannot tell autograd about its update and stays off."""
    missing = []
    bump = getattr(getattr(getattr(t, "autograd", None), "graph", None), "increment_version", None)
    if bump is None:
        missing.append("torch.autograd.graph.increment_version")
    foreach_add = getattr(t, "_foreach_add_", None)
    if foreach_add is None:
        missing.append("torch._foreach_add_")

        def add_one(tensors, value=1):       # the public spelling, one host call per counter
            for x in tensors:
                x.add_(value)
    else:
        add_one = foreach_add
    return bump, add_one, missing


_bump_versions, _add_to_each, missing = _probe()


def register(module, params):
    """Called by fused.py when a module's Parameters have been re-pointed at an engine's arena."""
    for p in params:
        _OWNER[id(p)] = (weakref.ref(p), weakref.ref(module))
    install()


def install():
    global _installed
    if _installed:
        return
    from torch.optim.optimizer import register_optimizer_step_post_hook, register_optimizer_step_pre_hook
    register_optimizer_step_pre_hook(_pre_step)
    register_optimizer_step_post_hook(_post_step)
    _installed = True


def _t_ptr(dev, t):
    ring = _T_RING.get(dev)
    if ring is None or t > ring.numel():
        n = 1 << 16
        while n < t:
            n <<= 1
        ring = _T_RING[dev] = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
    return ring.data_ptr() + 8 * (t - 1)


def _eligible(opt, group):
    if type(opt) is not torch.optim.Adam:
        return False
    if tuple(group.get("betas", ())) != (0.9, 0.999) or group.get("eps") != 1e-8 or group.get("weight_decay", 0) != 0:
        return False
    for flag in ("amsgrad", "maximize", "capturable", "differentiable", "fused", "decoupled_weight_decay"):
        if group.get(flag):
            return False
    return True


def _decline(why):
    stats["declined"][why] = stats["declined"].get(why, 0) + 1


_PLANS = weakref.WeakKeyDictionary()   # optimizer -> (signature of its groups, [(group index, weakref(module)), ...])


def _discover(opt):
    """Which (param group, VAE module) pairs of this optimizer the hook may take over: the group holds ALL ten
    parameters of a module whose Parameters live in an engine's arena.  Re-derived when the groups change.
    -> [(group index, weakref(module)), ...]"""
    from . import fused
    found = []
    for gi, group in enumerate(opt.param_groups):
        ids = {id(q) for q in group["params"]}
        seen = set()
        mods = []
        for p in group["params"]:
            own = _OWNER.get(id(p))
            if own is None or own[0]() is not p:
                continue
            module = own[1]()
            if module is None or id(module) in seen:
                continue
            seen.add(id(module))
            mine = fused._params(module)
            if any(id(q) not in ids for q in mine):
                _decline("group holds only part of the model")   # PyTorch's step
                continue
            mods.append(module)
        if len(mods) == 1:     # (two fused models in one group: left to PyTorch)
            found.append((gi, weakref.ref(mods[0])))
        elif mods:
            _decline("several fused models in one group")
    return found


def _restore(opt):
    stash = _STASH.pop(opt, None)
    if stash:
        for params, grads in stash:
            for q, g in zip(params, grads):
                q.grad = g


def _pre_step(opt, args, kwargs):
    # (`args` is the step call's positional arguments INCLUDING the optimizer itself)
    if not enabled or not _OWNER:
        return None
    # a stock step that raised never reached the post-hook: the gradients it left hidden were consumed by the fused
    # update of THAT step and must not come back under a later one
    _STASH.pop(opt, None)
    if _bump_versions is None:
        _decline("this torch lacks " + missing[0])
        return None
    if len(args) > 1 or kwargs.get("closure") is not None:
        return None
    from . import fused, ops
    groups = opt.param_groups
    sig = (len(_OWNER),) + tuple([len(g["params"]) for g in groups])
    plans = _PLANS.get(opt)
    if plans is None or plans[0] != sig:
        plans = _PLANS[opt] = (sig, _discover(opt))
    for gi, mref in plans[1]:
        group, module = groups[gi], mref()
        if module is None:
            continue
        if not _eligible(opt, group):
            _decline("optimizer type or hyper-parameters")
            continue
        holder = fused._HOLDERS.get(module)
        params = None if holder is None else holder.last_params     # the tensors the last fused forward ran on
        if params is None or not holder.engines:
            _decline("no engine")
            continue
        eng = holder.last_engine
        grads = [q.grad for q in params]
        ok = True
        f32 = torch.float32
        for g in grads:
            if g is None or g.dtype is not f32 or not g.is_cuda or g.is_sparse or not g.is_contiguous():
                ok = False
                break
        if not ok:
            _decline("gradients missing / not dense fp32 on the device")
            continue
        if not _fused_adam(opt, group, holder, eng, params, grads):
            continue
        stats["fused_steps"] += 1
        # the stock step that follows must not touch these parameters: it skips parameters without a gradient, so their
        # gradients are hidden until the post-hook.  The group's list stays whole whatever the stock step does.
        _STASH.setdefault(opt, []).append((params, grads))
        for q in params:
            q.grad = None
        # the kernel refreshed this engine's operand shadows; the per-layer Functions' caches (ops.py) are stale
        ops.invalidate_shadows()
        eng._shared["version"] += 1
        eng._shadow_version = eng._shared["version"]
        # an in-place update as far as autograd is concerned (what torch.optim's own kernels do to the version counters)
        _bump_versions(params)
        holder.versions = tuple([q._version for q in params]) + (ops._EPOCH[0],)
    return None


def _post_step(opt, args, kwargs):
    _restore(opt)


def _fused_adam(opt, group, holder, eng, params, grads):
    from .engine import PARAM_NAMES
    state = opt.state
    cache = holder.adam_cache
    if cache is None or cache["arena"] is not eng.exp_avg:
        # moments: views of the engine's arenas (every engine of a holder shares them), in torch's state layout
        cache = holder.adam_cache = {"arena": eng.exp_avg, "descs": {}, "steps": None,
                                     "m": [eng.view(eng.exp_avg, k) for k in PARAM_NAMES],
                                     "v": [eng.view(eng.exp_avg_sq, k) for k in PARAM_NAMES]}
    st0 = state[params[0]]
    if st0.get("exp_avg") is not cache["m"][0] or state[params[9]].get("exp_avg_sq") is not cache["v"][9]:
        # first step of this optimizer on these parameters, or state that a stock step / load_state_dict created
        for i, p in enumerate(params):
            st = state[p]
            if len(st) == 0:
                cache["m"][i].zero_()
                cache["v"][i].zero_()
                st["step"] = torch.tensor(0.0, dtype=torch.get_default_dtype())
            else:
                if st["exp_avg"] is not cache["m"][i]:
                    cache["m"][i].copy_(st["exp_avg"])
                if st["exp_avg_sq"] is not cache["v"][i]:
                    cache["v"][i].copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"] = cache["m"][i], cache["v"][i]
        cache["steps"] = [state[p]["step"] for p in params]
        st0 = state[params[0]]
    step_tensors = cache["steps"]
    if step_tensors is None or st0["step"] is not step_tensors[0]:      # (state replaced behind the moments' back)
        step_tensors = cache["steps"] = [state[p]["step"] for p in params]
    # The kernel applies ONE bias correction to all ten tensors.  A stock step in between that skipped some of them
    # (gradient None: the reference's loop never does that, a frozen layer would) leaves their counters behind; checked
    # on every call (ten host scalars), and such a step is PyTorch's.
    steps = [float(v) for v in step_tensors]
    if min(steps) != max(steps):
        _decline("parameters at different step counts")
        return False
    t = int(steps[0]) + 1
    descs = cache["descs"].get(eng.B)
    if descs is None:
        descs = (_lib.ParamDesc * 10)(*eng.plan_descs())
        for d in descs:            # gradient = one exact-shape tensor per parameter (what autograd left in .grad)
            d.grad_ld, d.grad_split_stride, d.grad_splits = d.cols, 0, 1
            d.grad_half, d.grad_unscale = 0, None
        cache["descs"][eng.B] = descs
    for i in range(10):
        descs[i].grad_slabs = grads[i].data_ptr()
    lr = group["lr"]
    lr = float(lr.item()) if torch.is_tensor(lr) else float(lr)
    lib().rv_adam_multi(descs, 10, ptr(eng.param), ptr(eng.exp_avg), ptr(eng.exp_avg_sq), None, None, lr, 1.0,
                        _t_ptr(eng.device, t), stream_ptr())
    _add_to_each(step_tensors, 1)
    return True
