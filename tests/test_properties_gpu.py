"""Size-independent properties checked at the benchmark's full size (C2: S=1024, H=2048, L=64,
B=4096), where the CPU oracle would be slow: bit-reproducibility, the data-parallel identity
(mean of half-batch gradients == full-batch gradient), masking of padded rows, loss descent,
and agreement of the two host paths (autograd API vs fused engine) on the same step."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle.inputs import PARAM_NAMES, make_eps, make_frames, make_params  # noqa: E402

S, H, L, B = 1024, 2048, 64, 4096
KL, LR = 1e-4, 1e-4


def _engine(batch, **kw):
    from rawaudiovae_kelsey_amd.engine import TrainEngine
    e = TrainEngine(S, H, L, batch, kl_beta=KL, lr=LR, **kw)
    e.load_params(make_params(S, H, L, 0))
    return e


def _grads(e, x, eps):
    from rawaudiovae_kelsey_amd import engine as E
    e.step(x, eps, phases=E.PHASE_FWD | E.PHASE_BWD_A | E.PHASE_BWD_B | E.PHASE_FINALIZE_A | E.PHASE_FINALIZE_B)
    torch.cuda.synchronize()
    return e.grad.clone(), e.last_loss()


def test_full_size_step_is_bit_reproducible():
    x = torch.from_numpy(make_frames(B, S, 1)).cuda()
    runs = []
    for _ in range(2):
        e = _engine(B, seed=11)
        for _ in range(3):
            e.step(x)
        torch.cuda.synchronize()
        runs.append((e.param.clone(), e.losses(3)))
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]


@pytest.mark.parametrize("slabs,bound", [("fp32", 2e-5), ("fp16", 5e-4)])
def test_data_parallel_identity_at_full_size(slabs, bound):
    """Two ranks with 2048 frames each: mean of their gradients == gradient of the 4096-frame batch
    (the reference's loss is a mean, rawvae/model.py:39,45): up to fp32 summation order with fp32 split-K slabs
    (2e-5), and up to the 11-bit significand of the partial sums with the default block-floating-point fp16 slabs
    (each partial is rounded once, 3e-4 expected, stated bound 5e-4)."""
    x = make_frames(B, S, 2)
    eps = make_eps(B, L, 3)
    xd, ed = torch.from_numpy(x).cuda(), torch.from_numpy(eps).cuda()
    g_full, l_full = _grads(_engine(B, slab_dtype=slabs), xd, ed)
    half = B // 2
    ga, la = _grads(_engine(half, slab_dtype=slabs), xd[:half].contiguous(), ed[:half].contiguous())
    gb, lb = _grads(_engine(half, slab_dtype=slabs), xd[half:].contiguous(), ed[half:].contiguous())
    mean = 0.5 * (ga + gb)
    rel = (mean - g_full).norm() / g_full.norm()
    assert rel < bound, float(rel)
    assert abs(0.5 * (la[0] + lb[0]) - l_full[0]) < 1e-6 * l_full[0]


def test_padded_rows_do_not_leak_into_gradients():
    """B = 4000 is padded to 4096 rows inside the engine; the result must equal the same 4000 frames
    run as part of nothing else, i.e. be independent of what the padding rows compute."""
    b = 4000
    x = torch.from_numpy(make_frames(b, S, 4)).cuda()
    eps = torch.from_numpy(make_eps(b, L, 5)).cuda()
    g1, l1 = _grads(_engine(b), x, eps)
    e2 = _engine(b)
    # poison the workspace (incl. padding rows of every activation buffer) before the step
    e2.workspace.fill_(0x7F)
    e2.refresh_shadows()
    g2, l2 = _grads(e2, x, eps)
    assert torch.isfinite(g2).all() and torch.equal(g1, g2) and l1 == l2


def test_loss_descends_and_moments_follow_adam():
    e = _engine(B, seed=3)
    xs = [torch.from_numpy(make_frames(B, S, 20 + i)).cuda() for i in range(4)]
    for i in range(40):
        e.step(xs[i % 4])
    ls = e.losses(40)
    assert ls[-1] < ls[0] and all(np.isfinite(ls))
    assert int(e.step_counter.item()) == 40
    # Adam's per-step |update| is bounded by lr*(1-b1)/sqrt(1-b2) ~ 3.2 lr and is ~lr for steady gradients
    p0 = np.concatenate([make_params(S, H, L, 0)[k].reshape(-1) for k in PARAM_NAMES])
    moved = np.abs(e.param.cpu().numpy() - p0).max()
    assert 0 < moved <= 40 * LR * 3.2


def test_api_path_and_engine_agree_on_one_step():
    """loss.backward() through the autograd Functions and the fused engine run the same kernels on
    the same operands: loss and gradients must agree to fp32 summation order."""
    from rawvae.model import VAE, loss_function
    b = 512
    x = torch.from_numpy(make_frames(b, S, 7)).cuda()
    eps = torch.from_numpy(make_eps(b, L, 8)).cuda()
    m = VAE(S, H, L)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(S, H, L, 0).items()})
    m = m.cuda()
    recon, mu, logvar = m(x, eps=eps)
    loss = loss_function(recon, x, mu, logvar, KL, S)
    loss.backward()
    g_api = torch.cat([p.grad.reshape(-1) for _, p in m.named_parameters()])
    g_eng, l_eng = _grads(_engine(b), x, eps)
    assert abs(loss.item() - l_eng[0]) < 2e-6 * l_eng[0]
    # the API path rounds d(pre-tanh) to bf16 from (2/N d)(1-r^2), the engine from 2/N (d (1-r^2)):
    # one-ulp fp32 differences that occasionally flip a bf16 rounding
    assert float((g_api - g_eng).norm() / g_eng.norm()) < 2e-4


def test_reference_default_ini_batch_131072():
    """The reference's own default.ini (default.ini:18-28): S=1024, H=2048, L=256, batch 131072 -- the
    largest size the path is configured for.  Checked through the data-parallel identity: the loss of
    the 131072-frame batch is the mean of the losses of its 32 chunks of 4096 frames, and its gradient
    is the mean of theirs (fp32 summation order only: this identity is checked on fp32 split-K slabs; the default
    fp16 slabs add the rounding of each partial sum, bounded in test_data_parallel_identity_at_full_size), all on the
    same weights and eps."""
    from rawaudiovae_kelsey_amd import engine as E
    from rawaudiovae_kelsey_amd.engine import TrainEngine
    Sd, Hd, Ld, Bd, chunk = 1024, 2048, 256, 131072, 4096
    p = make_params(Sd, Hd, Ld, 0)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(Bd, Sd, device="cuda", generator=g) * 2 - 1
    eps = torch.randn(Bd, Ld, device="cuda", generator=g)
    ph = E.PHASE_FWD | E.PHASE_BWD_A | E.PHASE_BWD_B | E.PHASE_FINALIZE_A | E.PHASE_FINALIZE_B
    big = TrainEngine(Sd, Hd, Ld, Bd, kl_beta=KL, lr=LR, slab_dtype="fp32")
    big.load_params(p)
    big.step(x, eps, phases=ph)
    torch.cuda.synchronize()
    g_big, l_big = big.grad.clone(), big.last_loss()
    mu_big = big.outputs()[0][-3:].clone()
    del big
    torch.cuda.empty_cache()
    small = TrainEngine(Sd, Hd, Ld, chunk, kl_beta=KL, lr=LR, slab_dtype="fp32")
    small.load_params(p)
    g_sum = torch.zeros_like(g_big, dtype=torch.float64)
    l_sum = np.zeros(3)
    for i in range(Bd // chunk):
        small.step(x[i * chunk:(i + 1) * chunk], eps[i * chunk:(i + 1) * chunk], phases=ph)
        torch.cuda.synchronize()
        g_sum += small.grad.double()
        l_sum += np.array(small.last_loss())
    n = Bd // chunk
    for a, b in zip(l_big, l_sum / n):
        assert abs(a - b) <= 2e-6 * abs(b), (l_big, l_sum / n)
    rel = float((g_big.double() - g_sum / n).norm() / (g_sum / n).norm())
    assert rel < 5e-5, rel
    # last rows of the last chunk: same GEMM inputs, only the split-K count (fp32 summation order) differs
    assert torch.allclose(mu_big, small.outputs()[0][-3:], rtol=0, atol=2e-6)


def test_soak_two_thousand_steps_bit_reproducible():
    """Race screen for the hand-placed LDS-DMA waits / barriers (ping-pong and ring main loops, dual
    launches): 2000 full-size steps on cycling batches, twice from the same state -- any tile read before
    its DMA landed, even once, shows up as a differing bit in the final parameters or the loss ring."""
    xs = [torch.from_numpy(make_frames(B, S, 50 + i)).cuda() for i in range(4)]
    st = torch.cuda.Stream()
    runs = []
    for _ in range(2):
        e = _engine(B, seed=21, ring=2048)
        with torch.cuda.stream(st):
            for i in range(2000):
                e.step(xs[i % 4], stream=st)
        st.synchronize()
        runs.append((e.param.clone(), e.exp_avg_sq.clone(), torch.tensor(e.losses(2000))))
        del e
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert torch.equal(runs[0][2], runs[1][2])
    assert bool(torch.isfinite(runs[0][2]).all()) and float(runs[0][2][-1]) < float(runs[0][2][0])
