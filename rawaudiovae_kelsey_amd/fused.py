"""`VAE.forward` under autograd as ONE graph node backed by a step plan.

The reference's training loop (train.py:184-193) is

    optimizer.zero_grad(); recon, mu, logvar = model(x)
    loss = loss_function(recon, x, mu, logvar, kl_beta, S); loss.backward(); optimizer.step()

Run through one autograd Function per layer group (ops.EncodeFn / ReparamFn / DecodeFn) that loop costs ~60
kernel launches and ~1 ms of host time per step at C2.  Here the forward is one host call -- the FWD phase of
`rv_plan_step` (cast, fc1, heads, reparam, fc3, fc4: the six kernels the fused engine runs) -- and the backward is
one more: `rv_plan_set_external_grads` hands the plan whatever gradients autograd delivers for (recon, mu, logvar)
-- any loss, not only `loss_function` -- and the BWD + FINALIZE phases produce all ten parameter gradients
(paired fc4 backward, latent pair, reparam backward, heads pair, fc1 weight gradient, one finalize launch).
The optimizer stays the caller's (`torch.optim.Adam` in the reference): the module's Parameters are re-pointed
at the plan's fp32 arena (`TrainEngine.adopt`, state_dict keys / layouts unchanged) and the bf16 operand shadows
are rebuilt when a Parameter's version counter has moved.

Limits (the per-layer Functions remain the general path and are used automatically otherwise): the input must
not require grad, every parameter must, and `backward` must run before the next fused `forward` of the same
batch size on the same module (activations live in the plan's workspace; a second forward overwrites them --
the error says so; set `model.fused_training = False` for such loops).
"""
import collections
import weakref

import torch

from . import _lib
from ._lib import (PHASE_BWD_A, PHASE_BWD_B, PHASE_FINALIZE_A, PHASE_FINALIZE_B, PHASE_FWD, lib, ptr, stream_ptr)
from .engine import PARAM_NAMES, TrainEngine

MAX_ENGINES = 4          # distinct batch sizes kept per module (each owns a workspace)
_HOLDERS = weakref.WeakKeyDictionary()   # module -> _Holder; off the module so that it pickles as before


def _params(module):
    # (through the modules' own dictionaries: `module.fc1.weight` costs two nn.Module.__getattr__ fallbacks per
    # parameter, ~20 us per call for the ten -- this runs on every forward of the drop-in loop)
    mods = module._modules
    out = []
    for name in ("fc1", "fc21", "fc22", "fc3", "fc4"):
        pd = mods[name]._parameters
        out.append(pd["weight"])
        out.append(pd["bias"])
    return tuple(out)


class _Holder:
    def __init__(self):
        self.engines = collections.OrderedDict()   # batch size -> TrainEngine (all share the first one's arenas)
        self.ptrs = None
        self.versions = None
        self.last_params = None     # the Parameters and the engine of the most recent fused forward (optim_hook.py)
        self.last_engine = None
        self.adam_cache = None

    def engine(self, module, B, params):
        eng = self.engines.get(B)
        dev = params[0].device
        if eng is not None and eng.device != dev:
            self.engines.clear()
            eng = None
        if eng is None:
            base = next(iter(self.engines.values()), None)
            eng = TrainEngine(module.segment_length, module.n_units, module.latent_dim, B, device=dev, kl_beta=0.0,
                              lr=0.0, seed=module._rng_seed, grad_arena=False, share=base)
            self.engines[B] = eng
            while len(self.engines) > MAX_ENGINES:
                self.engines.popitem(last=False)
        else:
            self.engines.move_to_end(B)
        ptrs = tuple(p.data_ptr() for p in params)
        if ptrs != self.ptrs:           # first use, or .to() / .cpu().cuda() / a swapped .data since
            eng.adopt(module)
            self.ptrs = tuple(p.data_ptr() for p in params)
            self.versions = None
            from . import optim_hook    # torch.optim.Adam.step() on these parameters as one fused launch
            optim_hook.register(module, params)
        from . import ops
        vers = tuple(p._version for p in params) + (ops._EPOCH[0],)
        if vers != self.versions:       # optimizer.step / load_state_dict / any in-place write / ops.invalidate_shadows()
            eng.params_changed()
            self.versions = vers
        self.last_params, self.last_engine = params, eng
        return eng


def fusable(module, x2):
    """The parameter tuple when `module(x2)` can run as one node on a step plan, else None."""
    if not getattr(module, "fused_training", True) or not torch.is_grad_enabled():
        return None
    if not x2.is_cuda or x2.requires_grad or x2.dtype != torch.float32 or x2.shape[0] == 0:
        return None
    if module.latent_dim > 256:
        return None
    params = _params(module)
    dev = x2.device
    for p in params:
        if not p.requires_grad or p.dtype != torch.float32 or p.device != dev:
            return None
    return params


class _FwdRecord:
    """What `loss_function` needs to recognise the untouched outputs of one fused forward (FusedLossFn below)."""
    __slots__ = ("eng", "tick", "eps", "x_ptr", "x_numel", "x_version", "params", "module", "loss_taken", "mu", "logvar",
                 "shortcut_consumed")


class VaeFn(torch.autograd.Function):
    """(recon, mu, logvar) = VAE.forward(x) -- rawvae/model.py:19-35 -- on a step plan."""

    @staticmethod
    def forward(ctx, x, eps, eng, rec, *params):
        B = x.shape[0]
        recon = torch.empty((B, eng.S), dtype=torch.float32, device=x.device)
        eng.step(x, eps=eps, recon_out=recon, phases=PHASE_FWD)
        mu, logvar = eng.outputs()
        ctx.eng, ctx.tick, ctx.eps, ctx.rec = eng, eng.host_steps, eps, rec
        rec.eng, rec.tick, rec.eps = eng, eng.host_steps, eps
        rec.x_ptr, rec.x_numel, rec.x_version = x.data_ptr(), x.numel(), x._version
        ctx.save_for_backward(recon)
        return recon, mu, logvar

    @staticmethod
    def backward(ctx, d_recon, d_mu, d_lv):
        eng = ctx.eng
        if ctx.rec.loss_taken:
            raise _lib.RvError(
                "rawvae fused forward: loss_function() took this forward's loss as ONE autograd node on the step plan "
                "(its backward runs the plan's own fused loss gradient), and now further gradients arrive for recon / mu / "
                "logvar from another term of the loss.  Set `model.fused_loss = False` to route loss_function through the "
                "general autograd path (any combination of losses), or add the extra term to the parameters only.")
        if eng.host_steps != ctx.tick:
            raise _lib.RvError(
                "rawvae fused forward: backward() reached a forward pass whose activations a later forward of the "
                "same batch size has overwritten. Call backward() before the next model(x), or set "
                "`model.fused_training = False` to use the per-layer autograd path.")
        (recon,) = ctx.saved_tensors
        dev = recon.device

        def f32c(t, shape):
            if t is None:
                return torch.zeros(shape, dtype=torch.float32, device=dev)
            return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()
        d_recon = f32c(d_recon, recon.shape)
        d_mu = None if d_mu is None else f32c(d_mu, None)
        d_lv = None if d_lv is None else f32c(d_lv, None)
        grad = torch.empty(eng.n_params, dtype=torch.float32, device=dev)
        L_ = lib()
        L_.rv_plan_set_external_grads(eng._plan, ptr(d_recon), ptr(recon), ptr(d_mu), ptr(d_lv), ptr(grad))
        try:
            L_.rv_plan_step(eng._plan, PHASE_BWD_A | PHASE_BWD_B | PHASE_FINALIZE_A | PHASE_FINALIZE_B, None,
                            ptr(ctx.eps), None, 0.0, 0.0, 1.0, 0, eng.seed, stream_ptr())
        finally:
            L_.rv_plan_set_external_grads(eng._plan, None, None, None, None, None)
        out = [None, None, None, None]
        need = ctx.needs_input_grad
        # (weights are 2-D views of their piece; a bias IS its piece: five view calls, not ten)
        for i, (piece, shape) in enumerate(zip(grad.split_with_sizes(eng.param_sizes), eng.param_shape_list)):
            out.append((piece.view(shape) if len(shape) > 1 else piece) if need[4 + i] else None)
        return tuple(out)


class FusedLossFn(torch.autograd.Function):
    """`loss_function(recon, x, mu, logvar, kl_beta, S)` (rawvae/model.py:38-47) on the UNTOUCHED outputs of a fused
    forward, as one autograd node over the model's parameters.

    The forward phase of the step plan has already done the loss's work: the fc4 GEMM's epilogue left the squared-error
    partial sums and dP4 = d(mse)/d(pre-tanh), the reparameterisation its KL partial sums.  So the value is one small
    launch (`rv_plan_loss`), and the backward is the plan's own backward -- paired fc4 backward on that dP4, latent and
    heads backward with the KL gradient folded in, fc1's weight gradient, one finalize launch -- exactly the kernels of
    the fused engine's step, with the upstream gradient of the scalar applied on the device in the finalize launch
    (`rv_plan_set_loss_grad`).  Against the general route (LossFn + VaeFn.backward over gradients from outside) this is
    one autograd node instead of two and five kernel launches fewer (the stand-alone loss kernel over recon / x, the
    scaling of its three gradients, tanh' and the bias column sums of dP4, a memset).  The graph edge from the loss to
    (recon, mu, logvar) does not exist on this route; VaeFn.backward refuses to run beside it (another loss term on the
    same outputs needs `model.fused_loss = False`)."""

    @staticmethod
    def forward(ctx, rec, kl_beta, *params):
        eng = rec.eng
        out = torch.empty(4, dtype=torch.float32, device=eng.device)
        lib().rv_plan_loss(eng._plan, float(kl_beta), ptr(out), stream_ptr())
        ctx.rec, ctx.kl = rec, float(kl_beta)
        rec.loss_taken = True
        return out[0]

    @staticmethod
    def run_backward(rec, kl_beta, g):
        """The plan's backward for the forward recorded in `rec`, times the device scalar `g`: the ten gradients as
        exact-shape views of one fresh flat tensor."""
        eng = rec.eng
        if eng.host_steps != rec.tick:
            raise _lib.RvError(
                "rawvae fused loss: backward() reached a forward pass whose activations a later forward of the same batch "
                "size has overwritten. Call backward() before the next model(x), or set `model.fused_training = False`.")
        g = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.contiguous().float()
        grad = torch.empty(eng.n_params, dtype=torch.float32, device=eng.device)
        L_ = lib()
        L_.rv_plan_set_loss_grad(eng._plan, ptr(g), ptr(grad))
        try:
            L_.rv_plan_step(eng._plan, PHASE_BWD_A | PHASE_BWD_B | PHASE_FINALIZE_A | PHASE_FINALIZE_B, None,
                            ptr(rec.eps), None, kl_beta, 0.0, 1.0, 0, eng.seed, stream_ptr())
        finally:
            L_.rv_plan_set_loss_grad(eng._plan, None, None)
        # (weights are 2-D views of their piece; a bias IS its piece: five view calls, not ten)
        return [piece.view(shape) if len(shape) > 1 else piece
                for piece, shape in zip(grad.split_with_sizes(eng.param_sizes), eng.param_shape_list)]

    @staticmethod
    def backward(ctx, g):
        pieces = FusedLossFn.run_backward(ctx.rec, ctx.kl, g)
        need = ctx.needs_input_grad
        return (None, None) + tuple(p if need[2 + i] else None for i, p in enumerate(pieces))


class FusedLoss(torch.Tensor):
    """The 0-dim loss tensor `fused_loss` hands back: an ordinary tensor in every respect (value, `.item()`, arithmetic
    -- results are plain tensors --, and its grad_fn is FusedLossFn's node, so `torch.autograd.backward`, `.grad()` and a
    `(2 * loss).backward()` all take the autograd engine's route), except that calling `.backward()` ON IT, the way the
    reference loop does (train.py:191), runs the node's backward right there on the calling thread and accumulates the
    ten gradients into `.grad` itself.  Same kernels, same gradients; what it skips is the autograd engine's hand-over to
    its device worker thread and back for a graph of one node (~80 us of the loop's ~140 us backward at C2,
    profiles/r05_api_breakdown.txt).  Anything the shortcut does not cover -- a `gradient` / `inputs` argument,
    `create_graph`, hooks on a parameter or on the loss, gradients disabled -- goes to `Tensor.backward` unchanged.
    LIMIT (round-5 advisor): hooks registered on the parameters' AccumulateGrad NODES (what torch's DistributedDataParallel
    reducer, Horovod and FSDP-style wrappers use) and global autograd hooks cannot be seen from here and never fire on the
    shortcut.  So it is taken only while no torch.distributed process group exists -- a gradient-synchronising wrapper
    needs one -- unless the module says `fused_backward_shortcut = True` explicitly; `= False` always takes the engine's
    route.  Like autograd, a second `.backward()` on the same loss raises unless the first passed `retain_graph=True`."""
    __torch_function__ = torch._C._disabled_torch_function_impl      # ops on it return plain tensors

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        rec = getattr(self, "_rv_rec", None)
        if (rec is None or gradient is not None or inputs is not None or create_graph or not torch.is_grad_enabled()
                or self._backward_hooks or not _plain_leaves(rec.params) or not _shortcut_allowed(rec)):
            return torch.Tensor.backward(self, gradient, retain_graph, create_graph, inputs)
        if getattr(rec, "shortcut_consumed", False):
            raise RuntimeError("Trying to backward through the graph a second time: the fused step's activations were "
                               "consumed by the first .backward(); pass retain_graph=True to the first call if a second "
                               "one is needed.")
        pieces = FusedLossFn.run_backward(rec, self._rv_kl, _one(rec.eng.device))
        if not retain_graph:
            rec.shortcut_consumed = True
        with torch.no_grad():
            for q, g in zip(rec.params, pieces):
                if q.grad is None:
                    q.grad = g
                else:
                    q.grad.add_(g)


_ONES = {}


def _one(dev):
    t = _ONES.get(dev)
    if t is None:
        t = _ONES[dev] = torch.ones((), dtype=torch.float32, device=dev)
    return t


def _shortcut_allowed(rec):
    """`module.fused_backward_shortcut`: True / False decide; unset (None): only while no torch.distributed process group
    exists (see FusedLoss)."""
    module = rec.module()
    flag = getattr(module, "fused_backward_shortcut", None) if module is not None else False
    if flag is not None:
        return bool(flag)
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized())


def _plain_leaves(params):
    """Leaf parameters without tensor hooks or post-accumulate hooks (those must see the engine's route)."""
    for q in params:
        if q._backward_hooks or getattr(q, "_post_accumulate_grad_hooks", None) or not q.requires_grad:
            return False
    return True


def fused_loss(recon_x, x, mu, logvar, kl_beta, segment_length):
    """The loss as FusedLossFn when (recon_x, mu, logvar) are the untouched outputs of the most recent fused forward of
    their model and `x` is the batch that forward read; None otherwise (the caller takes the general path)."""
    rec = getattr(recon_x, "_rv_fwd", None)
    if rec is None or rec.loss_taken or rec.mu() is not mu or rec.logvar() is not logvar or not torch.is_grad_enabled():
        return None
    if type(kl_beta) not in (float, int) or recon_x._version or mu._version or logvar._version:
        return None
    eng = rec.eng
    if eng.host_steps != rec.tick or segment_length != eng.S or not torch.is_tensor(x):
        return None
    if x.data_ptr() != rec.x_ptr or x.numel() != rec.x_numel or x._version != rec.x_version or x.requires_grad:
        return None
    module = rec.module()
    if module is None or not getattr(module, "fused_loss", True):
        return None
    out = FusedLossFn.apply(rec, kl_beta, *rec.params).as_subclass(FusedLoss)
    out._rv_rec, out._rv_kl = rec, float(kl_beta)
    return out


def forward(module, x2, eps=None, params=None):
    """`VAE.forward` body for a fusable call: x2 is [B, S] fp32 on the module's device (`params`: what `fusable`
    returned)."""
    holder = _HOLDERS.get(module)
    if holder is None:
        holder = _HOLDERS[module] = _Holder()
    if params is None:
        params = _params(module)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    if eps is not None:
        eps = eps.reshape(x2.shape[0], module.latent_dim)
        eps = eps if (eps.dtype == torch.float32 and eps.is_contiguous()) else eps.contiguous().float()
    eng = holder.engine(module, x2.shape[0], params)
    eng.seed = module._rng_seed
    rec = _FwdRecord()
    rec.params, rec.module, rec.loss_taken = params, weakref.ref(module), False
    recon, mu, logvar = VaeFn.apply(x2, eps, eng, rec, *params)
    # the three tensors the caller receives carry the record: loss_function recognises them by identity (fused_loss)
    # (weak references: the autograd node behind mu / logvar holds the record, so strong ones would close a cycle that
    # keeps the node's saved recon -- 16 MB at C2 -- alive until the garbage collector finds it)
    rec.mu, rec.logvar = weakref.ref(mu), weakref.ref(logvar)
    recon._rv_fwd = rec
    return recon, mu, logvar
