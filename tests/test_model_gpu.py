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
    """tutorial.ipynb:461,505-506,922-923 call encode / reparameterize / decode on their own."""
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
    c = O.forward(p, x, eps, quant="bf16")
    assert mu.shape == (B, L) and recon.shape == (B, S)
    np.testing.assert_allclose(mu.cpu().numpy(), c["mu"], atol=2e-3)
    np.testing.assert_allclose(logvar.cpu().numpy(), c["logvar"], atol=2e-3)
    np.testing.assert_allclose(recon.cpu().numpy(), c["recon"], atol=3e-3)
    assert torch.isfinite(z2).all() and not torch.equal(z2, z)


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
