"""Drop-in surface on the GPU: rawvae.model.VAE / loss_function used exactly as the
reference's train.py uses them (train.py:159-193), against the golden vectors."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import GOLDEN  # noqa: E402
from oracle import vae_oracle as O  # noqa: E402
from oracle.inputs import PARAM_NAMES, make_eps, make_frames, make_params  # noqa: E402


def _model(S, H, L):
    from rawvae.model import VAE
    m = VAE(S, H, L)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(S, H, L, 0).items()})
    return m.to("cuda")


def _rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-300))


def test_reference_loop_small_vs_golden():
    from rawvae.model import loss_function
    g = np.load(os.path.join(GOLDEN, "small_f32.npz"))
    S, H, L, B = (int(v) for v in g["shape"])
    m = _model(S, H, L)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    traj = []
    for i in range(20):
        x = torch.from_numpy(make_frames(B, S, 1234 + i)).cuda()
        eps = torch.from_numpy(make_eps(B, L, 4321 + i)).cuda()
        opt.zero_grad()
        recon, mu, logvar = m(x, eps=eps)
        loss = loss_function(recon, x, mu, logvar, 1e-4, S)
        loss.backward()
        if i == 0:
            assert loss.dim() == 0
            np.testing.assert_allclose(recon.detach().cpu().numpy(), g["recon"], atol=1e-2)
            np.testing.assert_allclose(mu.detach().cpu().numpy(), g["mu"], atol=1e-2)
            np.testing.assert_allclose(logvar.detach().cpu().numpy(), g["logvar"], atol=1e-2)
            grads = dict(m.named_parameters())
            p = O.cast_params(make_params(S, H, L, 0), np.float32)
            cq = O.forward(p, make_frames(B, S, 1234), make_eps(B, L, 4321), quant="bf16")
            gq = O.backward(p, cq, 1e-4, quant="bf16")
            for k in PARAM_NAMES:
                got = grads[k].grad.cpu().numpy()
                assert got.shape == g["grad/" + k].shape
                assert _rel_l2(got, g["grad/" + k].astype(np.float64)) < 6e-2, k
                assert _rel_l2(got, gq[k].astype(np.float64)) < 5e-3, k
        traj.append(loss.item())
        opt.step()
    rel = np.abs(np.array(traj) - g["traj"]) / g["traj"]
    assert rel.max() < 1e-3, rel


@pytest.mark.parametrize("B", [1, 37, 256])
def test_encode_decode_inference_calls(B):
    """tutorial.ipynb:461,505-506,922-923 call encode / reparameterize / decode on their own, under
    no_grad: exact-fp32 kernels, so the outputs match the fp32 reference arithmetic to summation order
    (1e-5; SURVEY 8d strict-fp32 gate), not merely to bf16 rounding."""
    S, H, L = 128, 256, 16
    m = _model(S, H, L)
    p = O.cast_params(make_params(S, H, L, 0), np.float32)
    x = make_frames(B, S, 5)
    eps = make_eps(B, L, 6)
    with torch.no_grad():
        mu, logvar = m.encode(torch.from_numpy(x).cuda())
        z = m.reparameterize(mu, logvar, eps=torch.from_numpy(eps).cuda())
        recon = m.decode(z)
        z2 = m.reparameterize(mu, logvar)
    c = O.forward(O.cast_params(p, np.float64), x.astype(np.float64), eps.astype(np.float64))
    assert mu.shape == (B, L) and recon.shape == (B, S)
    np.testing.assert_allclose(mu.cpu().numpy(), c["mu"], atol=1e-5)
    np.testing.assert_allclose(logvar.cpu().numpy(), c["logvar"], atol=1e-5)
    np.testing.assert_allclose(recon.cpu().numpy(), c["recon"], atol=1e-5)
    assert torch.isfinite(z2).all() and not torch.equal(z2, z)
    # the bf16 kernels stay selectable for inference
    m.inference_precision = "bf16"
    with torch.no_grad():
        recon_b = m(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).cuda())[0]
    cq = O.forward(p, x, eps, quant="bf16")
    np.testing.assert_allclose(recon_b.cpu().numpy(), cq["recon"], atol=3e-3)
    assert not torch.equal(recon_b, recon)


def test_inference_matches_reference_golden_fp32():
    """No-grad forward vs outputs captured from the reference itself (tests/golden/small_f32.npz)."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "small_f32.npz"))
    S, H, L, B = (int(v) for v in g["shape"])
    m = _model(S, H, L)
    x, eps = make_frames(B, S, 1234), make_eps(B, L, 4321)
    with torch.no_grad():
        recon, mu, logvar = m(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).cuda())
    for got, key in ((recon, "recon"), (mu, "mu"), (logvar, "logvar")):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (3, 5, 7), (64, 64, 16), (65, 63, 17), (130, 200, 1000), (256, 1024, 2048)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_fp32_kernel_any_shape(M, N, K, act):
    """rv_linear_fp32 on exact, unaligned shapes (incl. an offset view -> scalar-load path) vs float64."""
    from rawaudiovae_kelsey_amd import ops
    rng = np.random.default_rng(M * 131 + N * 17 + K)
    x = rng.uniform(-1, 1, (M, K + 1)).astype(np.float32)[:, 1:]        # not 16-byte aligned rows
    W = (rng.uniform(-1, 1, (N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.uniform(-1, 1, N).astype(np.float32)
    y = ops.linear_fp32(torch.from_numpy(np.ascontiguousarray(x)).cuda(), torch.from_numpy(W).cuda(),
                        torch.from_numpy(b).cuda(), act).cpu().numpy()
    ref = x.astype(np.float64) @ W.astype(np.float64).T + b
    ref = np.maximum(ref, 0) if act == 1 else np.tanh(ref) if act == 2 else ref
    np.testing.assert_allclose(y, ref, atol=3e-6)


def test_forward_accepts_1d_frame():
    """export-onnx.ipynb:361-362 feeds a single 1-D frame."""
    S, H, L = 128, 256, 16
    m = _model(S, H, L)
    with torch.no_grad():
        recon, mu, logvar = m(torch.from_numpy(make_frames(1, S, 3)[0]).cuda())
    assert recon.shape == (1, S) and mu.shape == (1, L) and logvar.shape == (1, L)


def test_loss_function_matches_oracle_and_backprops_upstream_scale():
    from rawvae.model import loss_function
    rng = np.random.default_rng(0)
    B, S, L = 48, 200, 12
    recon = torch.from_numpy(np.tanh(rng.standard_normal((B, S))).astype(np.float32)).cuda().requires_grad_()
    x = torch.from_numpy(rng.uniform(-1, 1, (B, S)).astype(np.float32)).cuda()
    mu = torch.from_numpy(rng.standard_normal((B, L)).astype(np.float32)).cuda().requires_grad_()
    lv = torch.from_numpy((0.2 * rng.standard_normal((B, L))).astype(np.float32)).cuda().requires_grad_()
    loss = loss_function(recon, x, mu, lv, 1e-4, S)
    ref, _, _ = O.loss_function(recon.detach().cpu().numpy().astype(np.float64), x.cpu().numpy().astype(np.float64),
                                mu.detach().cpu().numpy().astype(np.float64), lv.detach().cpu().numpy().astype(np.float64), 1e-4)
    assert abs(loss.item() - ref) < 1e-5 * ref
    (3.0 * loss).backward()
    np.testing.assert_allclose(recon.grad.cpu().numpy(), 3 * 2 * (recon.detach().cpu().numpy() - x.cpu().numpy()) / (B * S),
                               rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(mu.grad.cpu().numpy(), 3 * 1e-4 * mu.detach().cpu().numpy() / (B * L), rtol=1e-5, atol=1e-12)


def test_engine_adopts_module_parameters():
    """VAE.engine(): the fused step updates the same storage state_dict() reads."""
    S, H, L, B = 128, 256, 16, 64
    m = _model(S, H, L)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    eng = m.engine(B, kl_beta=1e-4, lr=1e-4, seed=1)
    x = torch.from_numpy(make_frames(B, S, 9)).cuda()
    eng.step(x)
    torch.cuda.synchronize()
    after = m.state_dict()
    assert list(after) == list(before)
    for k in before:
        d = (after[k] - before[k]).abs().max().item()
        assert 0 < d <= 1.01e-4, (k, d)
    with torch.no_grad():
        recon, mu, logvar = m(x)   # API path reads the updated weights
    assert torch.isfinite(recon).all()


def test_encoder_decoder_views_share_the_vae():
    """`Encoder(vae)` / `Decoder(vae)` (north_star's class surface; the reference has only VAE) own no
    parameters, leave the VAE's state_dict keys untouched and compute what `vae.encode` / `vae.decode` compute."""
    from rawvae.model import VAE, Decoder, Encoder
    m = VAE(64, 96, 8).cuda()
    keys = list(m.state_dict().keys())
    enc, dec = Encoder(m), Decoder(m)
    assert list(enc.parameters()) == [] and list(dec.parameters()) == [] and list(m.state_dict().keys()) == keys
    assert keys == ["fc1.weight", "fc1.bias", "fc21.weight", "fc21.bias", "fc22.weight", "fc22.bias",
                    "fc3.weight", "fc3.bias", "fc4.weight", "fc4.bias"]
    x = torch.rand(5, 64, device="cuda") * 2 - 1
    with torch.no_grad():
        mu, lv = enc(x)
        mu2, lv2 = m.encode(x)
        assert torch.equal(mu, mu2) and torch.equal(lv, lv2)
        assert torch.equal(dec(mu), m.decode(mu))


def _loop(m, steps, S, L, B, fused, loss_fn=None):
    from rawvae.model import loss_function
    m.fused_training = fused
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = []
    for i in range(steps):
        x = torch.from_numpy(make_frames(B, S, 50 + i)).cuda()
        eps = torch.from_numpy(make_eps(B, L, 90 + i)).cuda()
        opt.zero_grad()
        recon, mu, logvar = m(x, eps=eps)
        loss = loss_function(recon, x, mu, logvar, 1e-2, S) if loss_fn is None else loss_fn(recon, x, mu, logvar)
        loss.backward()
        losses.append(loss.item())
        if i == 0:
            first = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
            outs = [t.detach().clone() for t in (recon, mu, logvar)]
        opt.step()
    return losses, first, outs, {k: v.detach().clone() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("S,H,L,B", [(64, 96, 8, 16), (512, 256, 24, 300), (1024, 2048, 64, 4096)])
def test_fused_forward_node_equals_per_layer_functions(S, H, L, B):
    """VAE.forward as one autograd node on a step plan (fused.py) against the per-layer Functions
    (ops.EncodeFn / ReparamFn / DecodeFn): same outputs, same ten gradients, same parameters after torch.optim.Adam
    steps -- the reference's loop (train.py:184-193) unchanged in both cases.  The two paths run the same kernels
    on the same bf16 operands except where the fused plan pairs GEMMs in one launch (other split-K orders)."""
    la, ga, oa, pa = _loop(_model(S, H, L), 4, S, L, B, fused=True)
    lb, gb, ob, pb = _loop(_model(S, H, L), 4, S, L, B, fused=False)
    # mu / logvar: fp32 sums in another split-K order; recon: a last-bit difference in z may cross a bf16 rounding
    # boundary of the fc3 operand, which moves that row's reconstruction by ~1e-4
    for a, b in zip(oa[1:], ob[1:]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert _rel_l2(oa[0].cpu().numpy(), ob[0].double().cpu().numpy()) < 2e-4
    np.testing.assert_allclose(oa[0].cpu().numpy(), ob[0].cpu().numpy(), atol=5e-3, rtol=0)
    for k in PARAM_NAMES:
        assert ga[k].shape == gb[k].shape
        assert _rel_l2(ga[k].cpu().numpy(), gb[k].double().cpu().numpy()) < 2e-3, k
    np.testing.assert_allclose(la, lb, rtol=5e-5)
    for k in PARAM_NAMES:
        # Adam moves an element by ~lr whatever its gradient's size, so the few elements whose tiny gradient
        # changes sign between the two paths differ by up to 2 lr per step; all others agree closely
        d = (pa[k] - pb[k]).abs()
        assert float(d.max()) <= 4 * 2 * 1e-3 + 1e-6, k
        assert float(d.mean()) < 2e-5, k


def test_fused_forward_node_takes_any_loss_and_upstream_scale():
    """The node's backward consumes whatever autograd hands it: a hand-written loss built from torch ops on
    (recon, mu, logvar) -- not loss_function -- scaled by 3, against the per-layer path."""
    S, H, L, B = 128, 192, 16, 64

    def my_loss(recon, x, mu, logvar):
        return 3.0 * ((recon - x).abs().mean() + 0.1 * (mu * mu).mean() + 0.05 * logvar.exp().mean())
    la, ga, _, _ = _loop(_model(S, H, L), 2, S, L, B, fused=True, loss_fn=my_loss)
    lb, gb, _, _ = _loop(_model(S, H, L), 2, S, L, B, fused=False, loss_fn=my_loss)
    np.testing.assert_allclose(la, lb, rtol=2e-5)
    for k in PARAM_NAMES:
        assert _rel_l2(ga[k].cpu().numpy(), gb[k].double().cpu().numpy()) < 2e-3, k
    # a loss that ignores recon: its gradient arrives as None and must count as zero, not as the fused MSE gradient
    m = _model(S, H, L)
    x = torch.from_numpy(make_frames(B, S, 1)).cuda()
    recon, mu, logvar = m(x)
    (mu.sum() + logvar.sum()).backward()
    assert float(m.fc4.weight.grad.abs().max()) == 0.0 and float(m.fc3.bias.grad.abs().max()) == 0.0
    assert float(m.fc1.weight.grad.abs().max()) > 0.0


@pytest.mark.parametrize("S,H,L,B", [(64, 96, 8, 16), (512, 256, 24, 300), (1024, 2048, 64, 4096)])
def test_loss_function_on_fused_outputs_is_one_node_on_the_plan(S, H, L, B):
    """`loss_function` (model.py:38-47) called on the untouched outputs of the one-node forward -- the reference's loop,
    train.py:186-191 -- becomes ONE autograd node over the parameters whose backward is the step plan's own (fused
    engine kernels, the forward's fused MSE gradient; fused.FusedLossFn).  Against the general route (`fused_loss =
    False`: LossFn + VaeFn.backward on gradients from outside): the same loss to fp32 rounding of another summation order,
    the same ten gradients to the bf16 rounding of dP4 (computed from the fp32 accumulator on one route, from the fp32
    reconstruction on the other); against the fused ENGINE, which runs exactly these kernels: the same loss bit for bit.
    Upstream scale, fall-backs, and the guard against a second loss term on the same outputs."""
    from rawaudiovae_kelsey_amd import _lib
    from rawaudiovae_kelsey_amd.engine import TrainEngine
    from rawvae.model import loss_function
    x = torch.from_numpy(make_frames(B, S, 50)).cuda()
    eps = torch.from_numpy(make_eps(B, L, 90)).cuda()
    KLB = 1e-2

    def run(fused_loss, scale=1.0):
        m = _model(S, H, L)
        m.fused_loss = fused_loss
        recon, mu, logvar = m(x, eps=eps)
        loss = loss_function(recon, x, mu, logvar, KLB, S)
        (loss * scale if scale != 1.0 else loss).backward()
        return m, loss, {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    ma, la, ga = run(True)
    mb, lb, gb = run(False)
    from rawaudiovae_kelsey_amd import fused
    assert isinstance(la, fused.FusedLoss) and type(la.grad_fn.next_functions[0][0]).__name__ == "FusedLossFnBackward"
    assert type(lb.grad_fn).__name__ == "LossFnBackward" and type(la * 2.0) is torch.Tensor
    assert la.dim() == 0 and abs(la.item() - lb.item()) <= 5e-6 * abs(lb.item())
    for k in PARAM_NAMES:
        assert ga[k].shape == gb[k].shape == ma.state_dict()[k].shape
        assert _rel_l2(ga[k].cpu().numpy(), gb[k].double().cpu().numpy()) < 2e-3, k
    # the fused engine on the same weights, batch and eps: the same kernels, so the same loss bits and the same gradients
    e = TrainEngine(S, H, L, B, kl_beta=KLB, lr=1e-3)
    e.load_params(make_params(S, H, L, 0))
    from rawaudiovae_kelsey_amd import engine as E
    e.step(x, eps, phases=E.PHASE_FWD | E.PHASE_BWD_A | E.PHASE_BWD_B | E.PHASE_FINALIZE_A | E.PHASE_FINALIZE_B)
    assert e.last_loss()[0] == la.item()
    for k in PARAM_NAMES:
        assert torch.equal(e.grad_views()[k], ga[k]), k
    # `loss.backward()` above ran the node's backward on the calling thread (fused.FusedLoss.backward); through the autograd
    # engine -- torch.autograd.backward(loss), what any other caller gets -- the same bits; a second call accumulates
    m2 = _model(S, H, L)
    recon, mu, logvar = m2(x, eps=eps)
    l_eng = loss_function(recon, x, mu, logvar, KLB, S)
    torch.autograd.backward(l_eng, retain_graph=True)
    for k, p_ in m2.named_parameters():
        assert torch.equal(p_.grad, ga[k]), k
    l_eng.backward()
    for k, p_ in m2.named_parameters():
        assert torch.equal(p_.grad, 2.0 * ga[k]), k
    # a tensor hook on a parameter must see its gradient: the shortcut steps aside
    m3 = _model(S, H, L)
    seen = []
    m3.fc3.bias.register_hook(lambda g_: seen.append(g_.detach().clone()))
    recon, mu, logvar = m3(x, eps=eps)
    loss_function(recon, x, mu, logvar, KLB, S).backward()
    assert len(seen) == 1 and torch.equal(seen[0], ga["fc3.bias"]) and torch.equal(m3.fc1.weight.grad, ga["fc1.weight"])
    # the shortcut consumes the step's activations like autograd frees a graph: a second .backward() on the same loss raises
    # unless the first kept the graph; and `fused_backward_shortcut = False` sends .backward() through the autograd engine
    # (whose node hooks -- DistributedDataParallel's reducer, Horovod -- the shortcut cannot see): same bits
    m4 = _model(S, H, L)
    recon, mu, logvar = m4(x, eps=eps)
    l4 = loss_function(recon, x, mu, logvar, KLB, S)
    l4.backward()
    with pytest.raises(RuntimeError, match="second time"):
        l4.backward()
    recon, mu, logvar = m4(x, eps=eps)
    l4 = loss_function(recon, x, mu, logvar, KLB, S)
    for p_ in m4.parameters():
        p_.grad = None
    l4.backward(retain_graph=True)
    l4.backward()
    for k, p_ in m4.named_parameters():
        assert torch.equal(p_.grad, 2.0 * ga[k]), k
    m5 = _model(S, H, L)
    m5.fused_backward_shortcut = False
    fired = []
    recon, mu, logvar = m5(x, eps=eps)
    l5 = loss_function(recon, x, mu, logvar, KLB, S)
    node = l5.grad_fn.next_functions[0][0]
    node.register_hook(lambda gi, go: fired.append(1))        # a hook on an autograd NODE: only the engine's route fires it
    l5.backward()
    assert fired == [1]
    for k, p_ in m5.named_parameters():
        assert torch.equal(p_.grad, ga[k]), k
    # an upstream factor reaches the gradients on the device (no host read of it)
    _, _, g3 = run(True, scale=2.5)
    for k in PARAM_NAMES:
        np.testing.assert_allclose(g3[k].cpu().numpy(), 2.5 * ga[k].cpu().numpy(), rtol=2e-6, atol=0)
    # not the forward's batch, modified outputs, or a tensor kl_beta: the general route, same numbers
    m = _model(S, H, L)
    recon, mu, logvar = m(x, eps=eps)
    l2 = loss_function(recon, x.clone(), mu, logvar, KLB, S)
    assert type(l2.grad_fn).__name__ == "LossFnBackward" and abs(l2.item() - lb.item()) <= 5e-6 * abs(lb.item())
    recon, mu, logvar = m(x, eps=eps)
    l3 = loss_function(recon * 1.0, x, mu, logvar, KLB, S)
    assert type(l3.grad_fn).__name__ == "LossFnBackward"
    # a second loss term on the same outputs cannot be served beside the one-node loss: it says so, and names the switch
    recon, mu, logvar = m(x, eps=eps)
    l4 = loss_function(recon, x, mu, logvar, KLB, S) + 1e-3 * mu.pow(2).mean()
    with pytest.raises(_lib.RvError, match="fused_loss"):
        l4.backward()
    m.zero_grad()
    m.fused_loss = False
    recon, mu, logvar = m(x, eps=eps)
    (loss_function(recon, x, mu, logvar, KLB, S) + 1e-3 * mu.pow(2).mean()).backward()
    assert float(m.fc1.weight.grad.abs().max()) > 0.0
    # the record on the outputs does not keep the graph alive: nothing of a finished step is reachable from the model
    import gc
    import weakref
    m.fused_loss = True
    recon, mu, logvar = m(x, eps=eps)
    wr = weakref.ref(recon)
    loss = loss_function(recon, x, mu, logvar, KLB, S)
    loss.backward()
    del recon, mu, logvar, loss
    gc.disable()
    try:
        assert wr() is None        # freed by reference counting alone (no cycle through the record)
    finally:
        gc.enable()


def test_reference_loop_holds_no_memory_across_steps():
    """The drop-in loop hangs a record on the forward's outputs, returns a tensor subclass for the loss and hides /
    restores `.grad` fields around the stock optimizer step: none of it may keep a finished step's tensors alive.  300
    steps of the reference loop (train.py:184-193) at a mid-sized shape: device memory in use is the same after step 300 as
    after step 100 (the caching allocator's `memory_allocated` counts live tensors only), with the garbage collector off --
    reference counting alone must free everything."""
    import gc
    from rawvae.model import loss_function
    S, H, L, B = 512, 1024, 32, 512
    m = _model(S, H, L)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    xs = [torch.from_numpy(make_frames(B, S, 10 + i)).cuda() for i in range(4)]

    def step(i):
        x = xs[i % 4]
        opt.zero_grad()
        recon, mu, logvar = m(x)
        loss = loss_function(recon, x, mu, logvar, 1e-4, S)
        loss.backward()
        opt.step()
        return loss
    gc.collect()
    gc.disable()
    try:
        for i in range(100):
            step(i)
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        for i in range(100, 300):
            last = step(i)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() <= base + 4096, (torch.cuda.memory_allocated(), base)
        assert 0 < float(last.item()) < 1
    finally:
        gc.enable()


def test_fused_forward_node_guards_and_fallbacks():
    from rawaudiovae_kelsey_amd import _lib, fused
    S, H, L, B = 64, 96, 8, 16
    m = _model(S, H, L)
    x = torch.from_numpy(make_frames(B, S, 1)).cuda()
    r1, _, _ = m(x)
    r2, _, _ = m(x)                      # same batch size: overwrites the first forward's activations
    r2.sum().backward()
    with pytest.raises(_lib.RvError, match="fused_training"):
        r1.sum().backward()
    # other batch sizes get plans of their own that share the parameter arena; state_dict keys / shapes unchanged
    m.zero_grad()
    ra, _, _ = m(x)
    rb, _, _ = m(torch.from_numpy(make_frames(5, S, 2)).cuda())
    (ra.sum() + rb.sum()).backward()
    assert sorted(m.state_dict().keys()) == sorted(PARAM_NAMES)
    assert tuple(m.fc1.weight.shape) == (H, S) and m.fc1.weight.grad.shape == m.fc1.weight.shape
    # an input that requires grad, or a frozen parameter, takes the per-layer path
    assert not fused.fusable(m, x.clone().requires_grad_(True))
    m.fc3.bias.requires_grad_(False)
    assert not fused.fusable(m, x)
    recon, mu, logvar = m(x)
    recon.sum().backward()
    m.fc3.bias.requires_grad_(True)
    # load_state_dict / optimizer steps are seen through the parameters' version counters
    sd = {k: torch.from_numpy(v) for k, v in make_params(S, H, L, 3).items()}
    m.load_state_dict(sd)
    with torch.no_grad():
        want = m(x)[0]                   # exact-fp32 inference path on the new weights
    got = m(x, eps=None)[0]
    assert float((m.encode(x)[0] - m(x)[1]).detach().abs().max()) < 2e-2
    assert got.shape == want.shape
    # moving the module re-adopts its parameters
    m2 = m.cpu().cuda()
    r, mu, lv = m2(x)
    np.testing.assert_allclose(mu.detach().cpu().numpy(), m2.encode(x)[0].detach().cpu().numpy(), atol=2e-2)


def test_shadow_cache_is_tied_to_the_tensors_not_to_their_addresses():
    """The per-layer API path caches the padded bf16 / fp32 operand shadows of the parameters.  (1) A second model
    built after the first one was freed -- the allocator hands it the same addresses, with the same version
    counters -- must run on ITS weights (entries hold weak references to the tensors they were built from).
    (2) Writes through `.data` do not move a tensor's version counter: `ops.invalidate_shadows()` after them, and both
    the per-layer path and the one-node forward pick the new weights up."""
    import gc
    from rawvae.model import VAE
    from rawaudiovae_kelsey_amd import ops
    S, H, L, B = 128, 256, 8, 64
    x = torch.from_numpy(make_frames(B, S, 3)).cuda()

    def build(seed):
        m = VAE(S, H, L)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(S, H, L, seed).items()})
        m = m.to("cuda")
        m.fused_training = False        # per-layer Functions: the cached-shadow path
        return m
    outs, ptrs = [], []
    for seed in (0, 1):
        m = build(seed)
        ptrs.append(m.fc1.weight.data_ptr())
        mu, _ = m.encode(x)
        ref = O.forward(O.cast_params(make_params(S, H, L, seed), np.float32), x.cpu().numpy(),
                        np.zeros((B, L), np.float32), quant="bf16")
        np.testing.assert_allclose(mu.detach().cpu().numpy(), ref["mu"], atol=2e-4 * max(1, np.abs(ref["mu"]).max()))
        outs.append(mu.detach().clone())
        del m, mu
        gc.collect()
        torch.cuda.empty_cache()
    assert not torch.equal(outs[0], outs[1])
    # (2) a .data write is invisible to the version counters until the cache is invalidated
    for fused in (False, True):
        m = build(0)
        m.fused_training = fused
        a = m(x)[1].detach().clone()
        m.fc21.weight.data.mul_(2.0)
        m.fc21.bias.data.mul_(2.0)
        ops.invalidate_shadows()
        b = m(x)[1].detach().clone()
        np.testing.assert_allclose(b.cpu().numpy(), 2.0 * a.cpu().numpy(), rtol=2e-2, atol=1e-4)


def test_stock_adam_step_runs_as_one_fused_launch_and_keeps_torch_state():
    """optim_hook: `torch.optim.Adam(model.parameters()).step()` in the unchanged reference loop (train.py:163,193) is
    performed by rv_adam_multi -- one launch that also refreshes the operand shadows -- while optimizer.state keeps
    torch's layout.  Against the stock foreach step (hook off) from the same weights on the same batches: the same
    losses and parameters to the hardware sqrt / reciprocal (~3e-7 relative per update; an update is ~lr), the same
    moments; state_dict round trip; a changed lr is honoured; a group holding only part of the model is left alone."""
    from rawaudiovae_kelsey_amd import optim_hook
    from rawvae.model import loss_function
    S, H, L, B = 512, 256, 24, 300

    def loop(m, steps, hook, opt=None, first=0):
        optim_hook.enabled = hook
        opt = opt or torch.optim.Adam(m.parameters(), lr=1e-3)
        losses = []
        for i in range(first, first + steps):
            x = torch.from_numpy(make_frames(B, S, 50 + i)).cuda()
            eps = torch.from_numpy(make_eps(B, L, 90 + i)).cuda()
            opt.zero_grad()
            recon, mu, logvar = m(x, eps=eps)
            loss = loss_function(recon, x, mu, logvar, 1e-2, S)
            loss.backward()
            opt.step()
            assert m.fc1.weight.grad is not None      # gradients are visible again after step()
            losses.append(loss.item())
        return losses, opt
    try:
        import copy
        ma, mb = _model(S, H, L), _model(S, H, L)
        n0 = optim_hook.stats["fused_steps"]
        la, oa = loop(ma, 5, True)
        assert optim_hook.stats["fused_steps"] == n0 + 5, optim_hook.stats      # the hook did take the steps
        lb, ob = loop(mb, 5, False)
        assert optim_hook.stats["fused_steps"] == n0 + 5
        # The two updates agree to ~1e-7 relative per step (hardware sqrt / reciprocal); from the second forward on that
        # difference can move a weight across a bf16 rounding boundary of its operand shadow, which changes gradients at
        # the 1e-3 level, and Adam turns any gradient into a step of ~lr -- so: same loss trajectory to 2e-5, parameters
        # within a small fraction of lr on average (sign flips of near-zero gradients: up to 2 lr per step), moments alike
        np.testing.assert_allclose(la, lb, rtol=2e-5)
        for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
            d = (pa - pb).detach().abs()
            assert float(d.max()) <= 2.1 * 5 * 1e-3 and float(d.mean()) < 0.02 * 1e-3, (k, float(d.max()), float(d.mean()))
            sa, sb = oa.state[pa], ob.state[pb]
            assert float(sa["step"]) == float(sb["step"]) == 5.0
            # (five steps at lr 1e-3 apart, the two runs' gradients have drifted by a few per cent where they are small)
            assert _rel_l2(sa["exp_avg"].cpu().numpy(), sb["exp_avg"].double().cpu().numpy()) < 5e-2, k
            assert _rel_l2(sa["exp_avg_sq"].cpu().numpy(), sb["exp_avg_sq"].double().cpu().numpy()) < 5e-2, k
        # ONE update from identical state and gradients: the kernel against torch's own arithmetic, element by element
        x = torch.from_numpy(make_frames(B, S, 7)).cuda()
        eps = torch.from_numpy(make_eps(B, L, 8)).cuda()
        mf = _model(S, H, L)
        optim_hook.enabled = True
        of = torch.optim.Adam(mf.parameters(), lr=1e-3)
        recon, mu, logvar = mf(x, eps=eps)
        loss_function(recon, x, mu, logvar, 1e-2, S).backward()
        w0 = {k: p.detach().clone() for k, p in mf.named_parameters()}
        g0 = {k: p.grad.detach().clone() for k, p in mf.named_parameters()}
        of.step()
        for k, p in mf.named_parameters():
            g = g0[k].double()
            m_, v_ = 0.1 * g, 0.001 * g * g
            want = w0[k].double() - (1e-3 / 0.1) * m_ / ((v_.sqrt() / (0.001 ** 0.5)) + 1e-8)
            err = (p.detach().double() - want).abs().max()
            assert float(err) <= 1e-5 * 1e-3 + 4e-9, (k, float(err))     # an update of ~lr to 1e-5 of itself + fp32 rounding of w (|w| < 0.07)
        # checkpoint round trip (train.py:208-212): a fresh model + optimizer loaded from the state dicts continues alike
        mc = _model(S, H, L)
        mc.load_state_dict(ma.state_dict())
        oc = torch.optim.Adam(mc.parameters(), lr=1e-3)
        oc.load_state_dict(copy.deepcopy(oa.state_dict()))     # (as a checkpoint read from disk: no aliasing of ma's state)
        for g in oa.param_groups + oc.param_groups:
            g["lr"] = 5e-4                                                       # a scheduler's write
        la2, _ = loop(ma, 3, True, oa, first=5)
        lc2, _ = loop(mc, 3, True, oc, first=5)
        np.testing.assert_allclose(la2, lc2, rtol=1e-6)
        for pa, pc in zip(ma.parameters(), mc.parameters()):
            assert float((pa - pc).detach().abs().max()) <= 1e-7
        assert float(oc.state[mc.fc4.weight]["step"]) == 8.0
        # and against the stock step continuing from the same checkpoint with the same lr
        md = _model(S, H, L)
        md.load_state_dict(mb.state_dict())
        od = torch.optim.Adam(md.parameters(), lr=1e-3)
        od.load_state_dict(copy.deepcopy(ob.state_dict()))
        for g in od.param_groups:
            g["lr"] = 5e-4
        ld2, _ = loop(md, 3, False, od, first=5)
        np.testing.assert_allclose(la2, ld2, rtol=2e-5)      # (two runs 1e-7 apart per update, as above: 2e-5 on the losses)
        # a group with only some of the parameters: PyTorch's own step (state tensors are NOT arena views)
        me = _model(S, H, L)
        optim_hook.enabled = True
        oe = torch.optim.Adam([{"params": [me.fc1.weight, me.fc1.bias]}, {"params": [me.fc4.weight]}], lr=1e-3)
        x = torch.from_numpy(make_frames(B, S, 1)).cuda()
        recon, mu, logvar = me(x)
        loss_function(recon, x, mu, logvar, 1e-2, S).backward()
        before = me.fc3.weight.detach().clone()
        oe.step()
        assert torch.equal(me.fc3.weight, before) and float(oe.state[me.fc4.weight]["step"]) == 1.0
        # the next forward sees the stock step's in-place writes (version counters moved): shadows are rebuilt
        r2, _, _ = me(x)
        assert not torch.equal(r2, recon)
        # a group holding the ten parameters AND one more: the hook updates the ten, the stock step the other one; the
        # group's list is whole again afterwards, and the ten count as modified in place (version counters)
        mg = _model(S, H, L)
        extra = torch.nn.Parameter(torch.ones(7, device="cuda"))
        og = torch.optim.Adam(list(mg.parameters()) + [extra], lr=1e-3)
        recon, mu, logvar = mg(x)
        (loss_function(recon, x, mu, logvar, 1e-2, S) + extra.sum()).backward()
        v0 = mg.fc4.weight._version
        w0 = mg.fc4.weight.detach().clone()
        n0 = optim_hook.stats["fused_steps"]
        og.step()
        assert optim_hook.stats["fused_steps"] == n0 + 1
        assert len(og.param_groups[0]["params"]) == 11 and og.param_groups[0]["params"][-1] is extra
        assert mg.fc4.weight._version > v0 and not torch.equal(mg.fc4.weight, w0)
        assert og.state[mg.fc4.weight]["exp_avg"].data_ptr() != 0 and float(og.state[mg.fc4.weight]["step"]) == 1.0
        np.testing.assert_allclose(extra.detach().cpu().numpy(), 1.0 - 1e-3, rtol=1e-5)      # stock Adam's first step: -lr * sign(g)
        assert float(og.state[extra]["step"]) == 1.0
        og.zero_grad()
        assert mg.fc1.weight.grad is None and extra.grad is None
    finally:
        optim_hook.enabled = True
