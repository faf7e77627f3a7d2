"""Per-kernel parity tests (GPU): every C-ABI entry point against the CPU oracle /
numpy on the same seeded inputs.  bf16 operands are rounded identically on both
sides, so GEMM tolerances only cover fp32 accumulation-order differences."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import vae_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def L():
    from rawaudiovae_kelsey_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _lib.lib()


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def sp():
    return torch.cuda.current_stream().cuda_stream or None


def rand_bf16(rng, shape, scale=1.0):
    a = (rng.standard_normal(shape) * scale).astype(np.float32)
    return O.bf16_round(a)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 128, 192), (128, 256, 128), (64, 64, 64),
                                   (192, 64, 128), (64, 192, 256)])
def test_linear_fwd_bias_relu(L, M, N, K):
    rng = np.random.default_rng(M + N + K)
    x, w = rand_bf16(rng, (M, K)), rand_bf16(rng, (N, K))
    b = rng.standard_normal(N).astype(np.float32)
    xd, wd, bd = dev(x, torch.bfloat16), dev(w, torch.bfloat16), dev(b)
    y = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    for act in (0, 1):
        L.rv_linear_fwd(xd.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), M, N, K, act, y.data_ptr(), N, sp())
        ref = x.astype(np.float64) @ w.astype(np.float64).T + b
        if act:
            ref = np.maximum(ref, 0)
        got = y.float().cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-2 * np.abs(ref).max())
        # bf16 output: error must be within one bf16 ulp of the exact value
        assert np.abs(got - ref).max() <= 2 ** -7 * np.abs(ref).max()


@pytest.mark.parametrize("M,N,K,splits", [(128, 128, 256, 1), (128, 128, 256, 4), (64, 64, 128, 2),
                                          (256, 128, 128, 2)])
def test_linear_fwd_f32_splits(L, M, N, K, splits):
    rng = np.random.default_rng(7)
    x, w = rand_bf16(rng, (M, K)), rand_bf16(rng, (N, K))
    b = rng.standard_normal(N).astype(np.float32)
    xd, wd, bd = dev(x, torch.bfloat16), dev(w, torch.bfloat16), dev(b)
    y = torch.full((splits, M, N), 7.0, dtype=torch.float32, device="cuda")
    L.rv_linear_fwd_f32(xd.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), M, N, K, splits, y.data_ptr(), N, sp())
    got = y.sum(0).cpu().numpy()
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())


@pytest.mark.parametrize("M,N,K,splits", [(128, 128, 128, 1), (128, 256, 192, 1), (64, 64, 128, 2),
                                          (192, 64, 256, 4), (256, 128, 64, 1)])
def test_linear_dgrad_f32(L, M, N, K, splits):
    """dX[M,N] = dY[M,K] @ W[K,N] with W consumed in [out=K, in=N] layout (transposing LDS reads)."""
    rng = np.random.default_rng(11)
    dy, w = rand_bf16(rng, (M, K)), rand_bf16(rng, (K, N))
    dyd, wd = dev(dy, torch.bfloat16), dev(w, torch.bfloat16)
    out = torch.full((splits, M, N), 7.0, dtype=torch.float32, device="cuda")
    L.rv_linear_dgrad(dyd.data_ptr(), K, wd.data_ptr(), N, M, N, K, None, 0, None, 0, None,
                      out.data_ptr(), N, splits, sp())
    ref = dy.astype(np.float64) @ w.astype(np.float64)
    np.testing.assert_allclose(out.sum(0).cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())


@pytest.fixture(params=[-1, 0, 1, 2, 3, 4, 5, 7], ids=["auto", "t64", "t128w4", "t256x128w8", "t256x128w4", "t128w8", "t256x256", "t256x256pp"])
def tile(request, L):
    """Pin each block-tile configuration in turn (256-row tiles only apply when M % 256 == 0) through the test hook
    of include/rawvae_hip_diag.h."""
    t = request.param
    L.rv_gemm_force_tile(t)
    yield t
    L.rv_gemm_force_tile(-1)


@pytest.mark.parametrize("M,N,K,splits", [(256, 128, 256, 1), (512, 256, 192, 1), (256, 256, 512, 4)])
def test_all_tiles_all_layouts(L, tile, M, N, K, splits):
    """NT / NN / TN contractions under every tile configuration."""
    rng = np.random.default_rng(21)
    a_km, b_km = rand_bf16(rng, (M, K)), rand_bf16(rng, (N, K))
    b_mn, a_mn = rand_bf16(rng, (K, N)), rand_bf16(rng, (K, M))
    out = torch.empty((splits, M, N), dtype=torch.float32, device="cuda")
    A, Bk, Bm, Am = (dev(t, torch.bfloat16) for t in (a_km, b_km, b_mn, a_mn))
    L.rv_linear_fwd_f32(A.data_ptr(), K, Bk.data_ptr(), K, None, M, N, K, splits, out.data_ptr(), N, sp())
    ref = a_km.astype(np.float64) @ b_km.astype(np.float64).T
    np.testing.assert_allclose(out.sum(0).cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    L.rv_linear_dgrad(A.data_ptr(), K, Bm.data_ptr(), N, M, N, K, None, 0, None, 0, None, out.data_ptr(), N, splits, sp())
    ref = a_km.astype(np.float64) @ b_mn.astype(np.float64)
    np.testing.assert_allclose(out.sum(0).cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    L.rv_linear_wgrad(Am.data_ptr(), M, Bm.data_ptr(), N, M, N, K, splits, -1, out.data_ptr(), N, 0, None, sp())
    ref = a_mn.astype(np.float64).T @ b_mn.astype(np.float64)
    np.testing.assert_allclose(out.sum(0).cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())


@pytest.mark.parametrize("M,N,K", [(16384, 2048, 128), (16640, 2048, 256), (8192, 4096, 512), (33024, 1024, 256)])
def test_forward_gemm_tile_lists(L, M, N, K):
    """Large batches: a forward GEMM with at least 512 tiles of 256 x 256 runs as tile lists -- one workgroup per CU, the
    ping-pong loop staging across tile boundaries (gemm_pp_persist_kernel).  Same bits as the 128 x 128 tiles (the K order
    of every output element is the same), ragged list lengths included (520 and 516 tiles over 256 workgroups)."""
    rng = np.random.default_rng(31)
    a = dev(rand_bf16(rng, (M, K), 0.5), torch.bfloat16); w = dev(rand_bf16(rng, (N, K), 0.1), torch.bfloat16)
    bias = dev((rng.standard_normal(N) * 0.1).astype(np.float32))
    o1 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); o2 = torch.full_like(o1, 7.0)
    L.rv_linear_fwd(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), M, N, K, 1, o1.data_ptr(), N, sp())
    L.rv_gemm_force_tile(4)
    try:
        L.rv_linear_fwd(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), M, N, K, 1, o2.data_ptr(), N, sp())
    finally:
        L.rv_gemm_force_tile(-1)
    assert torch.equal(o1, o2)
    rows = rng.choice(M, 64, replace=False)
    ref = np.maximum(a[rows].float().cpu().numpy().astype(np.float64) @ w.float().cpu().numpy().astype(np.float64).T + bias.cpu().numpy(), 0)
    np.testing.assert_allclose(o1[rows].float().cpu().numpy(), ref, rtol=1e-2, atol=1e-3)


def test_all_tiles_fused_epilogues(L, tile):
    """bias+ReLU, mask+colsum and tanh+loss epilogues under every tile configuration."""
    from rawaudiovae_kelsey_amd._lib import gemm_tile
    M, N, K = 512, 256, 128
    rng = np.random.default_rng(22)
    a, w = rand_bf16(rng, (M, K)), rand_bf16(rng, (N, K), 0.1)
    wt = rand_bf16(rng, (K, N), 0.1)
    b = rng.standard_normal(N).astype(np.float32)
    A, W, WT, Bd = dev(a, torch.bfloat16), dev(w, torch.bfloat16), dev(wt, torch.bfloat16), dev(b)
    bm, bn = gemm_tile(M, N, 1)
    y = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    L.rv_linear_fwd(A.data_ptr(), K, W.data_ptr(), K, Bd.data_ptr(), M, N, K, 1, y.data_ptr(), N, sp())
    ref = np.maximum(a.astype(np.float64) @ w.astype(np.float64).T + b, 0)
    assert np.abs(y.float().cpu().numpy() - ref).max() <= 2 ** -7 * np.abs(ref).max()
    h = O.bf16_round(np.maximum(rng.standard_normal((M, N)), 0).astype(np.float32))
    Hd = dev(h, torch.bfloat16)
    cs = torch.zeros((M // bm, N), dtype=torch.float32, device="cuda")
    L.rv_linear_dgrad(A.data_ptr(), K, WT.data_ptr(), N, M, N, K, Hd.data_ptr(), N, y.data_ptr(), N,
                      cs.data_ptr(), None, 0, 1, sp())
    ref = (a.astype(np.float64) @ wt.astype(np.float64)) * (h > 0)
    assert np.abs(y.float().cpu().numpy() - ref).max() <= 2 ** -7 * np.abs(ref).max()
    np.testing.assert_allclose(cs.sum(0).cpu().numpy(), ref.sum(0), rtol=1e-4, atol=1e-4 * np.abs(ref.sum(0)).max())
    Bv, Sv = M - 37, N - 5   # ragged valid extents inside the padded tile grid
    x = rng.uniform(-1, 1, (Bv, Sv)).astype(np.float32)
    Xd = dev(x)
    recon = torch.zeros((Bv, Sv), device="cuda")
    dp4 = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    msep = torch.zeros((M // bm) * (N // bn), device="cuda")
    cs.zero_()
    L.rv_decode_out_loss_fwd(A.data_ptr(), K, W.data_ptr(), K, Bd.data_ptr(), M, N, K, Bv, Sv, Xd.data_ptr(), Sv,
                             recon.data_ptr(), Sv, dp4.data_ptr(), N, msep.data_ptr(), cs.data_ptr(), sp())
    rec = np.tanh(a.astype(np.float64) @ w.astype(np.float64).T + b)[:Bv, :Sv]
    np.testing.assert_allclose(recon.cpu().numpy(), rec, atol=2e-6)
    assert abs(msep.sum().item() - ((rec - x) ** 2).sum()) <= 1e-5 * ((rec - x) ** 2).sum()
    g = np.zeros((M, N))
    g[:Bv, :Sv] = 2.0 / (Bv * Sv) * (rec - x) * (1 - rec ** 2)
    got = dp4.float().cpu().numpy()
    assert np.abs(got - g).max() <= 2 ** -7 * np.abs(g).max()
    np.testing.assert_allclose(cs.sum(0).cpu().numpy(), g.sum(0), rtol=2e-3, atol=2e-3 * np.abs(g.sum(0)).max())


def test_linear_dgrad_mask_colsum(L):
    M, N, K = 256, 256, 128
    rng = np.random.default_rng(12)
    dy, w = rand_bf16(rng, (M, K)), rand_bf16(rng, (K, N))
    h = O.bf16_round(np.maximum(rng.standard_normal((M, N)), 0).astype(np.float32))
    dyd, wd, hd = dev(dy, torch.bfloat16), dev(w, torch.bfloat16), dev(h, torch.bfloat16)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    from rawaudiovae_kelsey_amd._lib import gemm_tile
    cs = torch.zeros((M // gemm_tile(M, N, 1)[0], N), dtype=torch.float32, device="cuda")
    L.rv_linear_dgrad(dyd.data_ptr(), K, wd.data_ptr(), N, M, N, K, hd.data_ptr(), N, out.data_ptr(), N,
                      cs.data_ptr(), None, 0, 1, sp())
    ref = (dy.astype(np.float64) @ w.astype(np.float64)) * (h > 0)
    got = out.float().cpu().numpy()
    assert np.abs(got - ref).max() <= 2 ** -7 * np.abs(ref).max()
    np.testing.assert_allclose(cs.sum(0).cpu().numpy(), ref.sum(0), rtol=1e-4, atol=1e-4 * np.abs(ref.sum(0)).max())


@pytest.mark.parametrize("M,N,K,splits", [(128, 128, 128, 1), (128, 256, 256, 2), (64, 64, 256, 4),
                                          (192, 64, 128, 1), (256, 128, 512, 4)])
def test_linear_wgrad(L, M, N, K, splits):
    """dW[M,N] = dY[K,M]^T @ X[K,N]; K is the batch."""
    rng = np.random.default_rng(13)
    dy, x = rand_bf16(rng, (K, M)), rand_bf16(rng, (K, N))
    dyd, xd = dev(dy, torch.bfloat16), dev(x, torch.bfloat16)
    out = torch.full((splits, M, N), 7.0, dtype=torch.float32, device="cuda")
    L.rv_linear_wgrad(dyd.data_ptr(), M, xd.data_ptr(), N, M, N, K, splits, -1, out.data_ptr(), N, 0, None, sp())
    ref = dy.astype(np.float64).T @ x.astype(np.float64)
    np.testing.assert_allclose(out.sum(0).cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())


def test_gemm_identity_asymmetric(L):
    """A = I against an asymmetric B catches transposed C writes / swapped fragment maps."""
    M = N = K = 128
    a = np.eye(M, dtype=np.float32)
    b = O.bf16_round((np.arange(K * N, dtype=np.float32).reshape(K, N) % 251) - 100)
    ad, bd = dev(a, torch.bfloat16), dev(b, torch.bfloat16)
    out = torch.zeros((1, M, N), dtype=torch.float32, device="cuda")
    L.rv_linear_dgrad(ad.data_ptr(), K, bd.data_ptr(), N, M, N, K, None, 0, None, 0, None, out.data_ptr(), N, 1, sp())
    np.testing.assert_array_equal(out[0].cpu().numpy(), b)
    L.rv_linear_wgrad(ad.data_ptr(), M, bd.data_ptr(), N, M, N, K, 1, -1, out.data_ptr(), N, 0, None, sp())
    np.testing.assert_array_equal(out[0].cpu().numpy(), b)
    bt = np.ascontiguousarray(b.T)
    btd = dev(bt, torch.bfloat16)
    L.rv_linear_fwd_f32(ad.data_ptr(), K, btd.data_ptr(), K, None, M, N, K, 1, out.data_ptr(), N, sp())
    np.testing.assert_array_equal(out[0].cpu().numpy(), b)


def test_cast_pad(L):
    rng = np.random.default_rng(3)
    for rows, cols, rp, cp in [(16, 64, 128, 128), (37, 100, 128, 128), (128, 256, 128, 256), (5, 7, 128, 128)]:
        a = rng.standard_normal((rows, cols)).astype(np.float32)
        ad = dev(a)
        out = torch.full((rp, cp), 9.0, dtype=torch.bfloat16, device="cuda")
        ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
        L.rv_cast_pad_bf16(ad.data_ptr(), rows, cols, cols, out.data_ptr(), rp, cp, cp, ctr.data_ptr(), sp())
        got = out.float().cpu().numpy()
        np.testing.assert_array_equal(got[:rows, :cols], O.bf16_round(a))
        assert np.all(got[rows:] == 0) and np.all(got[:, cols:] == 0)
        assert int(ctr.item()) == 1


def test_randn_statistics(L):
    from scipy import stats
    n = 1 << 20
    out = torch.zeros(n + 3, dtype=torch.float32, device="cuda")
    L.rv_randn(out.data_ptr(), n + 3, 1234, 5, sp())
    a = out.cpu().numpy().astype(np.float64)
    assert np.isfinite(a).all()
    assert abs(a.mean()) < 5e-3 and abs(a.std() - 1) < 5e-3
    assert abs(stats.skew(a)) < 2e-2 and abs(stats.kurtosis(a)) < 5e-2
    assert stats.kstest(a[:200000], "norm").pvalue > 1e-3
    out2 = torch.zeros_like(out)
    L.rv_randn(out2.data_ptr(), n + 3, 1234, 5, sp())
    assert torch.equal(out, out2)
    L.rv_randn(out2.data_ptr(), n + 3, 1234, 6, sp())
    assert not torch.equal(out, out2)
    assert abs(np.corrcoef(a, out2.cpu().numpy())[0, 1]) < 5e-3


@pytest.mark.parametrize("B,S,Lt", [(16, 64, 8), (32, 512, 8), (100, 1000, 3), (4096, 1024, 64)])
def test_loss_fused(L, B, S, Lt):
    rng = np.random.default_rng(5)
    recon = np.tanh(rng.standard_normal((B, S))).astype(np.float32)
    x = rng.uniform(-1, 1, (B, S)).astype(np.float32)
    mu = rng.standard_normal((B, Lt)).astype(np.float32)
    lv = (0.3 * rng.standard_normal((B, Lt))).astype(np.float32)
    kl_beta = 1e-4
    ws = torch.zeros(L.rv_loss_fused_workspace_bytes(), dtype=torch.uint8, device="cuda")
    rd, xd, md, ld = dev(recon), dev(x), dev(mu), dev(lv)
    out = torch.zeros(4, device="cuda")
    dr, dm, dl = torch.zeros_like(rd), torch.zeros_like(md), torch.zeros_like(ld)
    for _ in range(2):  # second call checks the ticket re-arms
        L.rv_loss_fused(rd.data_ptr(), xd.data_ptr(), md.data_ptr(), ld.data_ptr(), B, S, Lt, kl_beta,
                        out.data_ptr(), dr.data_ptr(), dm.data_ptr(), dl.data_ptr(), ws.data_ptr(), sp())
        loss, mse, kld = O.loss_function(recon.astype(np.float64), x.astype(np.float64), mu.astype(np.float64),
                                         lv.astype(np.float64), kl_beta)
        got = out.cpu().numpy()
        assert abs(got[0] - loss) <= 1e-5 * abs(loss)
        assert abs(got[1] - mse) <= 1e-5 * abs(mse) and abs(got[2] - kld) <= 1e-4 * abs(kld) + 1e-7
    np.testing.assert_allclose(dr.cpu().numpy(), 2 * (recon - x) / (B * S), rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(dm.cpu().numpy(), kl_beta * mu / (B * Lt), rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(dl.cpu().numpy(), kl_beta * 0.5 * (np.exp(lv) - 1) / (B * Lt), rtol=1e-4, atol=1e-12)


def test_reparameterize(L):
    rng = np.random.default_rng(6)
    n = 1000
    mu, lv, eps = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    z = torch.zeros(n, device="cuda")
    mud, lvd, epd = dev(mu), dev(lv), dev(eps)  # keep the device tensors alive across the calls
    L.rv_reparameterize(mud.data_ptr(), lvd.data_ptr(), n, epd.data_ptr(), None, 0, 0, z.data_ptr(), sp())
    np.testing.assert_allclose(z.cpu().numpy(), mu + eps * np.exp(0.5 * lv), rtol=2e-6, atol=1e-6)
    e_out = torch.zeros(n, device="cuda")
    L.rv_reparameterize(mud.data_ptr(), lvd.data_ptr(), n, None, e_out.data_ptr(), 9, 1, z.data_ptr(), sp())
    e = e_out.cpu().numpy()
    np.testing.assert_allclose(z.cpu().numpy(), mu + e * np.exp(0.5 * lv), rtol=2e-6, atol=1e-6)
    assert abs(e.mean()) < 0.15 and abs(e.std() - 1) < 0.15


@pytest.mark.parametrize("B,Lt,K,splits", [(100, 3, 256, 2), (4096, 64, 2048, 4), (256, 256, 128, 1)])
def test_heads_reparam_fwd(L, B, Lt, K, splits):
    """rv_heads_reparam_fwd (model.py:21-26 + the KL term of :45): fc21 | fc22 as one GEMM, z = mu + eps * std and the
    KL partials, against numpy on the same bf16-rounded operands; explicit eps and generated eps."""
    rng = np.random.default_rng(8)
    Bp, Lp = -(-B // 128) * 128, -(-Lt // 64) * 64
    h = np.zeros((Bp, K), np.float32); h[:B] = rand_bf16(rng, (B, K), 0.5)
    w = np.zeros((2 * Lp, K), np.float32)
    w[:Lt] = rand_bf16(rng, (Lt, K), 0.05); w[Lp:Lp + Lt] = rand_bf16(rng, (Lt, K), 0.05)
    bias = np.zeros(2 * Lp, np.float32)
    bias[:Lt] = rng.standard_normal(Lt) * 0.1; bias[Lp:Lp + Lt] = rng.standard_normal(Lt) * 0.1
    eps = rng.standard_normal((B, Lt)).astype(np.float32)
    hd, wd, bd, ed = dev(h, torch.bfloat16), dev(w, torch.bfloat16), dev(bias), dev(eps)
    slabs = torch.empty(splits, Bp, 2 * Lp, device="cuda")
    mulv = torch.empty(Bp, 2 * Lp, device="cuda")
    z = torch.empty(Bp, Lp, device="cuda", dtype=torch.bfloat16)
    klp = torch.zeros(Bp * Lp // 1024, device="cuda")
    ctr = torch.ones(1, dtype=torch.int64, device="cuda")
    L.rv_heads_reparam_fwd(hd.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), Bp, Lp, K, B, Lt, splits, slabs.data_ptr(),
                           ed.data_ptr(), None, 0, ctr.data_ptr(), mulv.data_ptr(), z.data_ptr(), klp.data_ptr(), sp())
    ref = h[:B].astype(np.float64) @ w.astype(np.float64).T + bias
    mu, lv = ref[:, :Lt], ref[:, Lp:Lp + Lt]
    got = mulv.cpu().numpy()
    np.testing.assert_allclose(got[:B, :Lt], mu, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(got[:B, Lp:Lp + Lt], lv, rtol=1e-5, atol=2e-5)
    assert not got[B:].any() and not got[:, Lt:Lp].any() and not got[:, Lp + Lt:].any()
    zr = O.bf16_round((mu + eps * np.exp(0.5 * lv)).astype(np.float32))
    zg = z.float().cpu().numpy()
    assert np.mean(zg[:B, :Lt] != zr) < 2e-3            # a last-bit fp32 difference may cross a bf16 rounding boundary
    np.testing.assert_allclose(zg[:B, :Lt], zr, rtol=1e-2, atol=1e-6)
    kl = float(klp.double().sum())
    kl_ref = float(np.sum(1 + lv - mu ** 2 - np.exp(lv)))
    assert abs(kl - kl_ref) <= 1e-5 * abs(kl_ref) + 1e-4
    # generated eps (Philox, written to eps_out) reproduces z through the same formula
    eo = torch.empty(B, Lt, device="cuda")
    L.rv_heads_reparam_fwd(hd.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), Bp, Lp, K, B, Lt, splits, slabs.data_ptr(),
                           None, eo.data_ptr(), 77, ctr.data_ptr(), mulv.data_ptr(), z.data_ptr(), klp.data_ptr(), sp())
    e2 = eo.cpu().numpy()
    assert abs(e2.mean()) < 0.05 + 2.0 / np.sqrt(e2.size) and abs(e2.std() - 1) < 0.1
    mv = mulv.cpu().numpy().astype(np.float64)
    z2 = O.bf16_round((mv[:B, :Lt] + e2 * np.exp(0.5 * mv[:B, Lp:Lp + Lt])).astype(np.float32))
    np.testing.assert_allclose(z.float().cpu().numpy()[:B, :Lt], z2, rtol=1e-2, atol=1e-6)


def _lp_of(Lt):
    return 64 if Lt <= 64 else 128 if Lt <= 128 else 256


@pytest.mark.parametrize("B,Lt,H", [(100, 3, 512), (4096, 64, 2048), (300, 40, 900), (200, 64, 1536),
                                    (4096, 256, 2048), (300, 100, 900), (200, 200, 640), (1000, 129, 1024),
                                    (16600, 256, 512), (33000, 100, 256), (8300, 64, 512)],   # large batches: 256-row tiles
                         ids=lambda v: str(v))
def test_latent_fwd_one_launch_equals_three(L, B, Lt, H):
    """rv_latent_fwd (heads GEMM + reparam + KL partials + fc3, model.py:21-29: one row-local launch at a padded latent
    width of 64, a GEMM with the reparameterisation in its epilogue + fc3's GEMM at 128 / 256 -- the reference's own
    latent_dim = 256, default.ini:18) against the route it replaces (rv_heads_reparam_fwd + rv_linear_fwd) and against
    numpy, on the same bf16 operands and eps."""
    rng = np.random.default_rng(9)
    Bp, Lp, Hp = -(-B // 128) * 128, _lp_of(Lt), -(-H // 512) * 512
    h = np.zeros((Bp, Hp), np.float32); h[:B, :H] = np.maximum(rand_bf16(rng, (B, H), 0.5), 0)
    wh = np.zeros((2 * Lp, Hp), np.float32)
    wh[:Lt, :H] = rand_bf16(rng, (Lt, H), 0.05); wh[Lp:Lp + Lt, :H] = rand_bf16(rng, (Lt, H), 0.05)
    bh = np.zeros(2 * Lp, np.float32)
    bh[:Lt] = rng.standard_normal(Lt) * 0.1; bh[Lp:Lp + Lt] = rng.standard_normal(Lt) * 0.1
    w3 = np.zeros((Hp, Lp), np.float32); w3[:H, :Lt] = rand_bf16(rng, (H, Lt), 0.2)
    b3 = np.zeros(Hp, np.float32); b3[:H] = rng.standard_normal(H) * 0.1
    eps = rng.standard_normal((B, Lt)).astype(np.float32)
    hd, whd, w3d = dev(h, torch.bfloat16), dev(wh, torch.bfloat16), dev(w3, torch.bfloat16)
    bhd, b3d, ed = dev(bh), dev(b3), dev(eps)
    ctr = torch.ones(1, dtype=torch.int64, device="cuda")

    def outs():
        return (torch.empty(Bp, 2 * Lp, device="cuda"), torch.empty(Bp, Lp, device="cuda", dtype=torch.bfloat16),
                torch.zeros(Bp * Lp // 1024, device="cuda"), torch.empty(Bp, Hp, device="cuda", dtype=torch.bfloat16))
    mulv1, z1, kl1, h31 = outs()
    L.rv_latent_fwd(hd.data_ptr(), Hp, whd.data_ptr(), Hp, bhd.data_ptr(), w3d.data_ptr(), Lp, b3d.data_ptr(), Bp, Hp, Lp,
                    B, Lt, ed.data_ptr(), None, 0, ctr.data_ptr(), mulv1.data_ptr(), z1.data_ptr(), kl1.data_ptr(),
                    h31.data_ptr(), Hp, sp())
    mulv3, z3, kl3, h33 = outs()
    splits = 2
    slabs = torch.empty(splits, Bp, 2 * Lp, device="cuda")
    L.rv_heads_reparam_fwd(hd.data_ptr(), Hp, whd.data_ptr(), Hp, bhd.data_ptr(), Bp, Lp, Hp, B, Lt, splits, slabs.data_ptr(),
                           ed.data_ptr(), None, 0, ctr.data_ptr(), mulv3.data_ptr(), z3.data_ptr(), kl3.data_ptr(), sp())
    L.rv_linear_fwd(z3.data_ptr(), Lp, w3d.data_ptr(), Lp, b3d.data_ptr(), Bp, Hp, Lp, 1, h33.data_ptr(), Hp, sp())
    ref = h[:B].astype(np.float64) @ wh.astype(np.float64).T + bh
    mu, lv = ref[:, :Lt], ref[:, Lp:Lp + Lt]
    got = mulv1.cpu().numpy()
    np.testing.assert_allclose(got[:B, :Lt], mu, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(got[:B, Lp:Lp + Lt], lv, rtol=1e-5, atol=2e-5)
    assert not got[B:].any() and not got[:, Lt:Lp].any() and not got[:, Lp + Lt:].any()
    np.testing.assert_allclose(got, mulv3.cpu().numpy(), rtol=1e-5, atol=2e-5)
    za, zb = z1.float().cpu().numpy(), z3.float().cpu().numpy()
    assert np.mean(za != zb) < 2e-3 and not za[B:].any() and not za[:, Lt:].any()
    np.testing.assert_allclose(za, zb, rtol=1e-2, atol=1e-6)
    kl_ref = float(np.sum(1 + lv - mu ** 2 - np.exp(lv)))
    assert abs(float(kl1.double().sum()) - kl_ref) <= 1e-5 * abs(kl_ref) + 1e-4
    if Lp == 64 and Bp <= 8192:   # (the GEMM forms keep one partial per tile and zeros in the slots beside it)
        np.testing.assert_allclose(kl1.cpu().numpy(), kl3.cpu().numpy(), rtol=1e-4, atol=1e-4)
    else:
        assert not kl1.cpu().numpy().reshape(-1, 4)[:, 1:].any()   # (the total was checked against float64 above)
    # fc3 on the kernel's own z (bf16) against float64: only the fp32 accumulation and the output rounding differ
    h3ref = np.maximum(za[:B].astype(np.float64) @ w3.astype(np.float64).T + b3, 0)
    np.testing.assert_allclose(h31.float().cpu().numpy()[:B], h3ref, rtol=1e-2, atol=1e-3)
    # where both routes rounded z alike, h3 differs only where another k order of the 64-term sum crosses a bf16
    # rounding boundary of the output (a few elements per million, one ulp)
    rows_same = (za == zb).all(axis=1)
    ha, hb = h31.float().cpu().numpy()[rows_same], h33.float().cpu().numpy()[rows_same]
    assert np.mean(ha != hb) < 1e-4
    np.testing.assert_allclose(ha, hb, rtol=1e-2, atol=1e-3)
    # generated eps: the same Philox draws as the three-launch route
    e1, e3 = torch.empty(B, Lt, device="cuda"), torch.empty(B, Lt, device="cuda")
    L.rv_latent_fwd(hd.data_ptr(), Hp, whd.data_ptr(), Hp, bhd.data_ptr(), w3d.data_ptr(), Lp, b3d.data_ptr(), Bp, Hp, Lp,
                    B, Lt, None, e1.data_ptr(), 77, ctr.data_ptr(), mulv1.data_ptr(), z1.data_ptr(), kl1.data_ptr(),
                    h31.data_ptr(), Hp, sp())
    L.rv_heads_reparam_fwd(hd.data_ptr(), Hp, whd.data_ptr(), Hp, bhd.data_ptr(), Bp, Lp, Hp, B, Lt, splits, slabs.data_ptr(),
                           None, e3.data_ptr(), 77, ctr.data_ptr(), mulv3.data_ptr(), z3.data_ptr(), kl3.data_ptr(), sp())
    assert torch.equal(e1, e3)
    from rawaudiovae_kelsey_amd import _lib
    with pytest.raises(_lib.RvError):
        L.rv_latent_fwd(hd.data_ptr(), Hp, whd.data_ptr(), Hp, bhd.data_ptr(), w3d.data_ptr(), Lp, b3d.data_ptr(), Bp, Hp, 512,
                        B, Lt, ed.data_ptr(), None, 0, ctr.data_ptr(), mulv1.data_ptr(), z1.data_ptr(), kl1.data_ptr(),
                        h31.data_ptr(), Hp, sp())


@pytest.mark.parametrize("B,Lt,H,ext", [(100, 3, 512, False), (4096, 64, 2048, False), (300, 40, 900, True), (200, 64, 1536, True),
                                        (4096, 256, 2048, False), (300, 100, 900, True), (200, 200, 640, True), (1000, 129, 1024, False),
                                        # large batches: dz and dW3 on 256 x 256 ping-pong tiles at Lp = 256, dz on 256 x 128 at 128
                                        (16600, 256, 512, True), (33000, 100, 256, False)])
def test_latent_bwd_one_launch_equals_two(L, B, Lt, H, ext):
    """rv_latent_bwd (dz = dP3 W3 over the full contraction + the reparameterisation / KL backward + the loss scalar
    + fc3's weight gradient on extra workgroups, one launch) against the route it replaces (rv_linear_dgrad into fp32 split-K slabs + rv_reparam_bwd) and against
    float64 numpy, on the same bf16 operands.  Stated bound: dz is a 512..2048-term bf16-product sum accumulated in
    fp32 in another order -> 1e-5 relative to the row's term scale; dmulv is rounded to bf16 once (<= 1 ulp apart)."""
    rng = np.random.default_rng(19)
    Bp, Lp, Hp, S = -(-B // 128) * 128, _lp_of(Lt), -(-H // 512) * 512, 512
    dp3 = np.zeros((Bp, Hp), np.float32); dp3[:B, :H] = rand_bf16(rng, (B, H), 1e-3)
    w3 = np.zeros((Hp, Lp), np.float32); w3[:H, :Lt] = rand_bf16(rng, (H, Lt), 0.2)
    mulv = np.zeros((Bp, 2 * Lp), np.float32)
    mulv[:B, :Lt] = rng.standard_normal((B, Lt)) * 0.5; mulv[:B, Lp:Lp + Lt] = rng.standard_normal((B, Lt)) * 0.3
    eps = rng.standard_normal((B, Lt)).astype(np.float32)
    xm = (rng.standard_normal((B, Lt)) * 1e-3).astype(np.float32) if ext else None
    xv = (rng.standard_normal((B, Lt)) * 1e-3).astype(np.float32) if ext else None
    n_mse, n_kl, kl_beta = 37, Bp // 16, 1e-2
    msep = rng.random(n_mse).astype(np.float32); klp = -rng.random(n_kl).astype(np.float32)
    dpd, w3d, mvd, ed = dev(dp3, torch.bfloat16), dev(w3, torch.bfloat16), dev(mulv), dev(eps)
    xmd, xvd = (dev(xm), dev(xv)) if ext else (None, None)
    msed, kld = dev(msep), dev(klp)
    ctr = torch.full((1,), 3, dtype=torch.int64, device="cuda")
    P = lambda t: None if t is None else t.data_ptr()
    zz = np.zeros((Bp, Lp), np.float32); zz[:B, :Lt] = rand_bf16(rng, (B, Lt), 1.0)
    zd = dev(zz, torch.bfloat16)
    w3s = 2 if Bp % 128 == 0 else 1
    dw3a, dw3b = torch.full((w3s, Hp, Lp), 7.0, device="cuda"), torch.empty(w3s, Hp, Lp, device="cuda")

    def outs():
        return (torch.full((Bp, 2 * Lp), 7.0, device="cuda", dtype=torch.bfloat16), torch.zeros(Bp // 16, 2 * Lp, device="cuda"),
                torch.zeros(4, 4, device="cuda"))
    dm1, db1, loss1 = outs()
    L.rv_latent_bwd(dpd.data_ptr(), Hp, w3d.data_ptr(), Lp, Bp, Hp, Lp, B, Lt, S, mvd.data_ptr(), ed.data_ptr(), kl_beta,
                    P(xmd), P(xvd), dm1.data_ptr(), db1.data_ptr(), msed.data_ptr(), n_mse, kld.data_ptr(), n_kl,
                    loss1.data_ptr(), ctr.data_ptr(), 4, zd.data_ptr(), Lp, dw3a.data_ptr(), Lp, w3s, sp())
    dm2, db2, loss2 = outs()
    splits = 4
    slabs = torch.empty(splits, Bp, Lp, device="cuda")
    L.rv_linear_dgrad(dpd.data_ptr(), Hp, w3d.data_ptr(), Lp, Bp, Lp, Hp, None, 0, None, 0, None, slabs.data_ptr(), Lp, splits, sp())
    L.rv_reparam_bwd(slabs.data_ptr(), splits, Bp, Lp, B, Lt, S, mvd.data_ptr(), ed.data_ptr(), kl_beta, P(xmd), P(xvd),
                     dm2.data_ptr(), db2.data_ptr(), msed.data_ptr(), n_mse, kld.data_ptr(), n_kl, loss2.data_ptr(),
                     ctr.data_ptr(), 4, sp())
    # float64 reference
    dz = dp3[:B].astype(np.float64) @ w3[:, :Lt].astype(np.float64)
    mu, lv = mulv[:B, :Lt].astype(np.float64), mulv[:B, Lp:Lp + Lt].astype(np.float64)
    sd = np.exp(0.5 * lv); ink = 1.0 / (B * Lt)
    dmu = dz + kl_beta * mu * ink + (xm if ext else 0)
    dlv = dz * eps * 0.5 * sd + kl_beta * 0.5 * (sd * sd - 1) * ink + (xv if ext else 0)
    g1, g2 = dm1.float().cpu().numpy(), dm2.float().cpu().numpy()
    scale = np.abs(dp3[:B]).astype(np.float64) @ np.abs(w3[:, :Lt]).astype(np.float64)
    tol = 2.0 ** -8 * np.abs(dmu) + 1e-5 * scale + 1e-12   # one bf16 rounding + the fp32 accumulation
    assert (np.abs(g1[:B, :Lt] - dmu) <= tol).all()
    tol_v = 2.0 ** -8 * np.abs(dlv) + 1e-5 * scale * np.abs(eps) * sd + 1e-12
    assert (np.abs(g1[:B, Lp:Lp + Lt] - dlv) <= tol_v).all()
    assert not g1[B:].any() and not g1[:, Lt:Lp].any() and not g1[:, Lp + Lt:].any()
    # against the two-launch route: equal up to one bf16 ulp where the fp32 sums differ in the last bits
    assert np.mean(g1 != g2) < 0.05
    np.testing.assert_allclose(g1, g2, rtol=2.0 ** -7, atol=1e-5 * float(scale.max()))   # (sums that nearly cancel)
    # bias-gradient partials: column sums of the UNROUNDED dmu / dlv over each 16-row block
    ref_db = np.zeros((Bp, 2 * Lp)); ref_db[:B, :Lt] = dmu; ref_db[:B, Lp:Lp + Lt] = dlv
    ref_db = ref_db.reshape(Bp // 16, 16, 2 * Lp).sum(1)
    d1, d2 = db1.cpu().numpy(), db2.cpu().numpy()
    if Lp > 64:   # the GEMM form: one partial row per 64-row tile in row 4 t of the table, zeros in rows 4 t + 1 .. + 3
        g = 16 if (Bp // 64) * (Lp // 64) > 1024 and Bp % 256 == 0 else 4   # (large batches: 256-row tiles, row 16 t)
        pad = np.zeros((-(-Bp // (16 * g)) * g, 2 * Lp))
        pad[:Bp // 16] = d1
        assert not pad.reshape(-1, g, 2 * Lp)[:, 1:].any()
        fold = lambda a: np.concatenate([a, np.zeros((-a.shape[0] % g, 2 * Lp))]).reshape(-1, g, 2 * Lp).sum(1)
        d1, d2, ref_db = fold(d1), fold(d2), fold(ref_db)
    np.testing.assert_allclose(d1, ref_db, rtol=1e-4, atol=1e-4 * np.abs(ref_db).max())
    np.testing.assert_allclose(d1, d2, rtol=1e-4, atol=1e-4 * np.abs(ref_db).max())
    # fc3's weight gradient on the launch's extra workgroups == rv_linear_wgrad (bit for bit where that picks the same
    # 64 x 64 tiles: N = 64), and right
    L.rv_linear_wgrad(dpd.data_ptr(), Hp, zd.data_ptr(), Lp, Hp, Lp, Bp, w3s, -1, dw3b.data_ptr(), Lp, 0, None, sp())
    if Lp == 64:
        assert torch.equal(dw3a, dw3b)
    ref_w3 = dp3.astype(np.float64).T @ zz.astype(np.float64)
    np.testing.assert_allclose(dw3a.double().sum(0).cpu().numpy(), ref_w3, rtol=1e-4, atol=1e-5 * np.abs(ref_w3).max())
    # the loss scalar: same summation order as rv_reparam_bwd's -> bit-equal; slot (3 - 1) % 4 of the ring
    assert torch.equal(loss1, loss2) and float(loss1[2, 0]) != 0.0 and not loss1[[0, 1, 3]].any()
    from rawaudiovae_kelsey_amd import _lib
    with pytest.raises(_lib.RvError):
        L.rv_latent_bwd(dpd.data_ptr(), Hp, w3d.data_ptr(), Lp, Bp, Hp, 512, B, Lt, S, mvd.data_ptr(), ed.data_ptr(), kl_beta,
                        None, None, dm1.data_ptr(), db1.data_ptr(), None, 0, None, 0, None, None, 0, None, 0, None, 0, 0, sp())


@pytest.mark.parametrize("B,Lt,H", [(512, 64, 128), (4096, 64, 2048), (1000, 17, 900)])
def test_heads_bwd_streaming_equals_the_dual_launch(L, B, Lt, H):
    """rv_heads_bwd (dP1 = relu'(h1) * (dmulv Wh), its column sums and dWh = dmulv^T h1 from ONE pass over h1) against
    the route it replaces (rv_linear_dgrad_wgrad: dgrad with mask + split-K wgrad) and against float64 numpy, on the
    same bf16 operands.  dP1: the K = 128 dot product is accumulated by the same four MFMAs in the same order -> bit
    equal; dWh and the column sums group the batch differently (512-row groups instead of K splits / 128-row tiles) ->
    equal to fp32 summation order (stated bound 1e-5 of the term scale)."""
    rng = np.random.default_rng(23)
    Bp, Lp, Hp = -(-B // 512) * 512, 64, -(-H // 128) * 128
    G = Bp // 512
    dm = np.zeros((Bp, 2 * Lp), np.float32)
    dm[:B, :Lt] = rand_bf16(rng, (B, Lt), 1e-3); dm[:B, Lp:Lp + Lt] = rand_bf16(rng, (B, Lt), 1e-3)
    wh = np.zeros((2 * Lp, Hp), np.float32)
    wh[:Lt, :H] = rand_bf16(rng, (Lt, H), 0.05); wh[Lp:Lp + Lt, :H] = rand_bf16(rng, (Lt, H), 0.05)
    h1 = np.zeros((Bp, Hp), np.float32); h1[:B, :H] = np.maximum(rand_bf16(rng, (B, H), 0.5), 0)
    dmd, whd, h1d = dev(dm, torch.bfloat16), dev(wh, torch.bfloat16), dev(h1, torch.bfloat16)
    dp1 = torch.full((Bp, Hp), 7.0, device="cuda", dtype=torch.bfloat16)
    db1 = torch.full((G, Hp), 7.0, device="cuda")
    dwh = torch.full((G, 2 * Lp, Hp), 7.0, device="cuda")
    L.rv_heads_bwd(dmd.data_ptr(), whd.data_ptr(), Hp, h1d.data_ptr(), Hp, Bp, Hp, Lp, dp1.data_ptr(), Hp, db1.data_ptr(),
                   dwh.data_ptr(), Hp, sp())
    # the generic route
    from rawaudiovae_kelsey_amd._lib import dgrad_wgrad_pick
    paired, bm, splits = dgrad_wgrad_pick(Bp, Hp, 2 * Lp)
    dp1r = torch.empty(Bp, Hp, device="cuda", dtype=torch.bfloat16)
    csr = torch.zeros(Bp // bm, Hp, device="cuda")
    dwr = torch.empty(splits, 2 * Lp, Hp, device="cuda")
    L.rv_linear_dgrad_wgrad(dmd.data_ptr(), 2 * Lp, whd.data_ptr(), Hp, h1d.data_ptr(), Hp, Bp, Hp, 2 * Lp, dp1r.data_ptr(), Hp,
                            csr.data_ptr(), dwr.data_ptr(), Hp, splits, 0, None, sp())
    assert torch.equal(dp1, dp1r)
    # float64
    ref = (dm.astype(np.float64) @ wh.astype(np.float64)) * (h1 > 0)
    got = dp1.float().cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=2.0 ** -7, atol=1e-5 * float(np.abs(ref).max()))
    assert not got[B:].any() and not got[:, H:].any()
    refw = dm.astype(np.float64).T @ h1.astype(np.float64)
    scale_w = float((np.abs(dm).astype(np.float64).T @ np.abs(h1).astype(np.float64)).max())
    np.testing.assert_allclose(dwh.double().sum(0).cpu().numpy(), refw, rtol=1e-4, atol=1e-5 * scale_w)
    np.testing.assert_allclose(dwh.double().sum(0).cpu().numpy(), dwr.double().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-5 * scale_w)
    refc = got.astype(np.float64).sum(0)            # column sums are taken of the fp32 values BEFORE the bf16 rounding:
    refc32 = ref.sum(0)                             # compare with the unrounded reference
    np.testing.assert_allclose(db1.double().sum(0).cpu().numpy(), refc32, rtol=1e-4, atol=1e-5 * float(np.abs(ref).sum(0).max()) + 1e-12)
    np.testing.assert_allclose(db1.double().sum(0).cpu().numpy(), csr.double().sum(0).cpu().numpy(), rtol=1e-4,
                               atol=1e-5 * float(np.abs(ref).sum(0).max()) + 1e-12)
    from rawaudiovae_kelsey_amd import _lib
    with pytest.raises(_lib.RvError):
        L.rv_heads_bwd(dmd.data_ptr(), whd.data_ptr(), Hp, h1d.data_ptr(), Hp, Bp, Hp, 128, dp1.data_ptr(), Hp, db1.data_ptr(),
                       dwh.data_ptr(), Hp, sp())


def test_adam_multi_and_finalize(L):
    from rawaudiovae_kelsey_amd._lib import ParamDesc
    rng = np.random.default_rng(8)
    shapes = [(5, 7), (1, 11), (3, 4)]
    n = sum(r * c for r, c in shapes)
    param = rng.standard_normal(n).astype(np.float32)
    m0 = (0.01 * rng.standard_normal(n)).astype(np.float32)
    v0 = (1e-4 * rng.random(n)).astype(np.float32)
    pd, md, vd = dev(param), dev(m0), dev(v0)
    gout = torch.zeros(n, device="cuda")
    descs = (ParamDesc * len(shapes))()
    keep, grads, off = [], [], 0
    shadows = []
    for i, (r, c) in enumerate(shapes):
        splits, ld = 3, c + 5
        slab = rng.standard_normal((splits, r + 2, ld)).astype(np.float32)
        sd = dev(slab)
        keep.append(sd)
        grads.append(slab[:, :r, :c].sum(0))
        sh = torch.zeros((r + 1, c + 3), dtype=torch.bfloat16, device="cuda")
        shadows.append(sh)
        descs[i] = ParamDesc(off, r, c, sd.data_ptr(), ld, (r + 2) * ld, splits, sh.data_ptr(), None, c + 3)
        off += r * c
    ctr = torch.full((1,), 3, dtype=torch.int64, device="cuda")
    L.rv_grad_finalize(descs, len(shapes), gout.data_ptr(), 0, sp())
    gflat = np.concatenate([g.reshape(-1) for g in grads])
    np.testing.assert_allclose(gout.cpu().numpy(), gflat, rtol=1e-6, atol=1e-6)
    L.rv_adam_multi(descs, len(shapes), pd.data_ptr(), md.data_ptr(), vd.data_ptr(), None, None, 1e-3, 0.5,
                    ctr.data_ptr(), sp())
    g = 0.5 * gflat.astype(np.float64)
    m = 0.9 * m0 + 0.1 * g
    v = 0.999 * v0 + 0.001 * g * g
    t = 3
    p = param - (1e-3 / (1 - 0.9 ** t)) * m / (np.sqrt(v) / np.sqrt(1 - 0.999 ** t) + 1e-8)
    np.testing.assert_allclose(md.cpu().numpy(), m, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(vd.cpu().numpy(), v, rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(pd.cpu().numpy(), p, rtol=1e-5, atol=1e-6)
    o = 0
    for (r, c), sh in zip(shapes, shadows):
        got = sh.float().cpu().numpy()
        np.testing.assert_array_equal(got[:r, :c], O.bf16_round(pd.cpu().numpy()[o:o + r * c].reshape(r, c)))
        assert np.all(got[r:] == 0) and np.all(got[:, c:] == 0)
        o += r * c


@pytest.mark.parametrize("loader", ["lds_dma_ring", "plain_loads"])
def test_wgrad_launch_carrying_optimizer_blocks(L, loader):
    """rv_linear_wgrad_adam: the weight-gradient GEMM must equal rv_linear_wgrad bit for bit, and the optimizer
    blocks riding in its launch must equal rv_adam_multi bit for bit on a mixed bag of tensors -- aligned matrices
    with 4, 2 and 1 split-K slabs (the streamed path), a ragged matrix, short rows, a bias row summed by a wave
    (>= 16 partials), a bias row with few partials -- for both loaders of the optimizer blocks: a bag of fp32 slabs
    only rides on the per-wave LDS-DMA rings, a bag with an fp16-slab tensor in it on the plain-load walk."""
    from rawaudiovae_kelsey_amd._lib import ParamDesc
    rng = np.random.default_rng(12)
    M, N, K, splits = 512, 256, 512, 2
    dy = dev(rand_bf16(rng, (K, M), 0.1), torch.bfloat16)
    x = dev(rand_bf16(rng, (K, N), 0.5), torch.bfloat16)
    # (rows, cols, splits, ld, fp16)
    spec = [(300, 256, 4, 256, False), (64, 128, 2, 136, False), (40, 64, 1, 64, False), (50, 37, 3, 44, False),
            (7, 3, 2, 4, False), (1, 200, 32, 208, False), (1, 77, 3, 80, False)]
    if loader == "plain_loads":
        spec.append((96, 64, 4, 64, True))
    n = sum(r * c for r, c, *_ in spec)
    n_pad = n + 16
    param = rng.standard_normal(n_pad).astype(np.float32)
    m0 = (0.01 * rng.standard_normal(n_pad)).astype(np.float32)
    v0 = (1e-4 * rng.random(n_pad)).astype(np.float32)
    descs = (ParamDesc * len(spec))()
    keep, off = [], 0
    shadows = []
    for i, (r, c, s, ld, half) in enumerate(spec):
        slab = (rng.standard_normal((s, r + 1, ld)) * 1e-2).astype(np.float32)
        us = None
        if half:
            # block-floating-point fp16 slabs: one power-of-two factor per 32 x 32 granule and slab (exponents
            # spread over 20 binades), values stored as fp16(partial * 2^e), the table holds 2^-e
            gr, gc = -(-(r + 1) // 32), ld // 32
            ex = rng.integers(-12, 9, size=(s, gr, gc))
            scale = np.repeat(np.repeat(np.exp2(ex), 32, axis=1), 32, axis=2)[:, :r + 1, :ld]
            sd = dev((slab * 4096 * scale).astype(np.float16))
            us = dev((np.exp2(-ex) / 4096).astype(np.float32))
        else:
            sd = dev(slab)
        keep.append((sd, us))
        cp = -(-c // 8) * 8
        sh = torch.zeros((r, cp), dtype=torch.bfloat16, device="cuda") if r > 1 else None
        shf = torch.zeros((1, cp), dtype=torch.float32, device="cuda") if r == 1 else None
        shadows.append((sh, shf))
        # offsets: multiples of 4 for the aligned tensors, odd ones in between stay valid for the scalar path
        descs[i] = ParamDesc(off, r, c, sd.data_ptr(), ld, (r + 1) * ld, s, sh.data_ptr() if sh is not None else None,
                             shf.data_ptr() if shf is not None else None, cp, None, None, int(half),
                             us.data_ptr() if half else None, ld // 32, (-(-(r + 1) // 32)) * (ld // 32))
        off += r * c
    ctr = torch.full((1,), 4, dtype=torch.int64, device="cuda")

    def state():
        return dev(param.copy()), dev(m0.copy()), dev(v0.copy())
    try:
        pa, ma, va = state()
        dwa = torch.zeros(splits, M, N, device="cuda")
        L.rv_linear_wgrad_adam(dy.data_ptr(), M, x.data_ptr(), N, M, N, K, splits, dwa.data_ptr(), N, 0, None, descs, len(spec),
                               pa.data_ptr(), ma.data_ptr(), va.data_ptr(), 1e-3, 0.25, ctr.data_ptr(), 24, sp())
        sha = [(a.clone() if a is not None else None, b.clone() if b is not None else None) for a, b in shadows]
        for a, b in shadows:
            for t in (a, b):
                if t is not None:
                    t.zero_()
        pb, mb, vb = state()
        dwb = torch.zeros(splits, M, N, device="cuda")
        L.rv_linear_wgrad(dy.data_ptr(), M, x.data_ptr(), N, M, N, K, splits, 7, dwb.data_ptr(), N, 0, None, sp())
        L.rv_adam_multi(descs, len(spec), pb.data_ptr(), mb.data_ptr(), vb.data_ptr(), None, None, 1e-3, 0.25, ctr.data_ptr(), sp())
        torch.cuda.synchronize()
    finally:
        pass
    assert torch.equal(dwa, dwb)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert not torch.equal(pa, dev(param))
    for (a1, b1), (a2, b2) in zip(sha, shadows):
        for t1, t2 in ((a1, a2), (b1, b2)):
            if t1 is not None:
                assert torch.equal(t1, t2)


@pytest.mark.parametrize("tile", [7, 2, 4, 0], ids=["256x256", "256x128", "128x128", "64x64"])
@pytest.mark.parametrize("mag", [1e-6, 1.0, 1e3, "mixed"])
def test_fp16_slabs_keep_their_precision_at_any_gradient_magnitude(L, tile, mag):
    """RV_SLAB_F16 is block floating point: every wave tile of every split-K slab carries its own power-of-two
    scale, so the fp16 partials of a weight gradient of ANY magnitude -- 1e-6, 1e+3, or 128-row blocks (the
    largest wave tile) 12 decades apart in one matrix -- sum (rv_grad_finalize) to the fp32-slab result within 1e-3 relative L2 (stated bound; 3e-4
    expected from fp16's 11-bit significand), block by block; a static scale would flush the small ones to zero or
    overflow the large ones.  Checked on every block tile a weight gradient can get."""
    from rawaudiovae_kelsey_amd._lib import ParamDesc
    rng = np.random.default_rng(77)
    M, N, K, splits = 512, 256, 512, 2
    dyh = rand_bf16(rng, (K, M), 0.1)
    if mag == "mixed":
        dyh = dyh * np.repeat(10.0 ** rng.uniform(-8, 4, M // 128), 128)[None, :]
    else:
        dyh = dyh * mag
    dyh = dyh.astype(np.float32)
    dy = dev(dyh, torch.bfloat16)
    x = dev(rand_bf16(rng, (K, N), 0.5), torch.bfloat16)
    w32 = torch.zeros(splits, M, N, device="cuda")
    w16 = torch.zeros(splits, M, N, dtype=torch.float16, device="cuda")
    us = torch.zeros(splits, M // 32, N // 32, device="cuda")
    L.rv_linear_wgrad(dy.data_ptr(), M, x.data_ptr(), N, M, N, K, splits, tile, w32.data_ptr(), N, 0, None, sp())
    L.rv_linear_wgrad(dy.data_ptr(), M, x.data_ptr(), N, M, N, K, splits, tile, w16.data_ptr(), N, 1, us.data_ptr(), sp())
    torch.cuda.synchronize()
    ush = us.cpu().numpy()
    assert np.all(ush > 0) and np.all(np.log2(ush) == np.round(np.log2(ush)))   # exact powers of two
    stored = w16.float().abs().amax().item()
    assert 2.0 ** 13 <= stored < 2.0 ** 15     # the largest tile maximum sits in fp16's top binades, never overflows
    deq = w16.float().cpu().numpy() * np.repeat(np.repeat(ush, 32, axis=1), 32, axis=2)
    ref = w32.cpu().numpy()
    for r0 in range(0, M, 128):    # every 128-row block on its own: the small blocks must be as good as the large ones
        a, b = deq[:, r0:r0 + 128], ref[:, r0:r0 + 128]
        assert np.linalg.norm(a - b) <= 1e-3 * np.linalg.norm(b), (r0, np.linalg.norm(a - b) / np.linalg.norm(b))
    # the optimizer's reader: rv_grad_finalize on a descriptor of the fp16 slabs == sum of the fp32 slabs
    out16 = torch.zeros(M * N, device="cuda")
    out32 = torch.zeros(M * N, device="cuda")
    d16 = (ParamDesc * 1)(ParamDesc(0, M, N, w16.data_ptr(), N, M * N, splits, None, None, 0, None, None, 1,
                                    us.data_ptr(), N // 32, (M // 32) * (N // 32)))
    d32 = (ParamDesc * 1)(ParamDesc(0, M, N, w32.data_ptr(), N, M * N, splits, None, None, 0))
    L.rv_grad_finalize(d16, 1, out16.data_ptr(), 0, sp())
    L.rv_grad_finalize(d32, 1, out32.data_ptr(), 0, sp())
    torch.cuda.synchronize()
    a, b = out16.view(M, N).cpu().numpy(), out32.view(M, N).cpu().numpy()
    np.testing.assert_array_equal(a, deq.sum(0, dtype=np.float32))
    for r0 in range(0, M, 128):
        assert np.linalg.norm(a[r0:r0 + 128] - b[r0:r0 + 128]) <= 1e-3 * np.linalg.norm(b[r0:r0 + 128]), r0


def test_gather_frames_matches_audio_dataset_semantics(L):
    """rv_gather_frames == AudioDataset.__getitem__ (rawvae/dataset.py:108-118) for a shuffled index,
    and == TestDataset for hop == S."""
    rng = np.random.default_rng(31)
    audio = rng.uniform(-1, 1, 5003).astype(np.float32)
    S, hop = 256, 64
    ref = O.hop_frames(audio, S, hop)
    n, padded = O.frame_count(len(audio), S, hop)
    buf = np.zeros(padded, np.float32)
    buf[:len(audio)] = audio
    ad = dev(buf)
    idx = rng.permutation(n).astype(np.int64)
    idd = dev(idx)
    out = torch.zeros((n, S), device="cuda")
    L.rv_gather_frames(ad.data_ptr(), padded, idd.data_ptr(), 0, n, S, hop, out.data_ptr(), sp())
    np.testing.assert_array_equal(out.cpu().numpy(), ref[idx])
    L.rv_gather_frames(ad.data_ptr(), padded, None, 3, n - 3, S, hop, out.data_ptr(), sp())
    np.testing.assert_array_equal(out.cpu().numpy()[:n - 3], ref[3:])
    ev = O.eval_frames(audio, 100)   # S not a multiple of 4: scalar tail path
    e = np.zeros(ev.size, np.float32)
    e[:len(audio)] = audio
    ed = dev(e)
    out2 = torch.zeros(ev.shape, device="cuda")
    L.rv_gather_frames(ed.data_ptr(), ev.size, None, 0, ev.shape[0], 100, 100, out2.data_ptr(), sp())
    np.testing.assert_array_equal(out2.cpu().numpy(), ev)


@pytest.mark.parametrize("force,loop", [(5, 102), (5, 108), (-1, 102)])
@pytest.mark.parametrize("M,N,K", [(512, 256, 256), (1024, 512, 256), (768, 256, 512), (1024, 768, 1024)])
def test_paired_dgrad_wgrad(L, force, loop, M, N, K):
    """rv_linear_dgrad_wgrad: dX = relu'(dY W) and dW = dY^T X from ONE launch (256x256 tiles, forced)
    and through its unpaired fallback; M = batch, K = out features, N = in features."""
    from rawaudiovae_kelsey_amd._lib import dgrad_wgrad_pick
    L.rv_gemm_force_tile(force)
    L.rv_gemm_force_tile(loop)       # 102: two-slot ring main loop, 108: ping-pong main loop
    try:
        rng = np.random.default_rng(41)
        dy, w = rand_bf16(rng, (M, K)), rand_bf16(rng, (K, N), 0.1)
        x = O.bf16_round(np.maximum(rng.standard_normal((M, N)), 0).astype(np.float32))
        paired, bm, splits = dgrad_wgrad_pick(M, N, K)
        assert paired == (1 if force == 5 else 0)
        DY, W, X = dev(dy, torch.bfloat16), dev(w, torch.bfloat16), dev(x, torch.bfloat16)
        dx = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
        cs = torch.zeros((M // bm, N), dtype=torch.float32, device="cuda")
        dw = torch.full((splits, K, N), 7.0, dtype=torch.float32, device="cuda")
        L.rv_linear_dgrad_wgrad(DY.data_ptr(), K, W.data_ptr(), N, X.data_ptr(), N, M, N, K, dx.data_ptr(), N,
                                cs.data_ptr(), dw.data_ptr(), N, splits, 0, None, sp())
        ref_dx = (dy.astype(np.float64) @ w.astype(np.float64)) * (x > 0)
        ref_dw = dy.astype(np.float64).T @ x.astype(np.float64)
        assert np.abs(dx.float().cpu().numpy() - ref_dx).max() <= 2 ** -7 * np.abs(ref_dx).max()
        np.testing.assert_allclose(cs.sum(0).cpu().numpy(), ref_dx.sum(0), rtol=1e-4, atol=1e-4 * np.abs(ref_dx.sum(0)).max())
        np.testing.assert_allclose(dw.sum(0).cpu().numpy(), ref_dw, rtol=1e-5, atol=1e-5 * np.abs(ref_dw).max())
    finally:
        L.rv_gemm_force_tile(-1)
        L.rv_gemm_force_tile(108)    # the default


@pytest.mark.parametrize("M,N,K,rounds", [(8192, 2048, 512, 2), (8192, 1024, 1024, 1), (12288, 2048, 512, 3), (8192, 2048, 1024, 2)])
def test_paired_dgrad_wgrad_several_rounds(L, M, N, K, rounds):
    """The picker pairs dgrad + wgrad into one 256 x 256 launch while their blocks come to one round of the 256 CUs or two to
    four rounds filled to three quarters (rv_dgrad_wgrad_pick; B = 8192 / 16384 in the batch sweep): the paired launch against
    float64 at those sizes, the block count checked from the picker's own numbers."""
    from rawaudiovae_kelsey_amd._lib import dgrad_wgrad_pick
    rng = np.random.default_rng(47)
    dy, w = rand_bf16(rng, (M, K), 0.1), rand_bf16(rng, (K, N), 0.1)
    x = O.bf16_round(np.maximum(rng.standard_normal((M, N)), 0).astype(np.float32))
    paired, bm, splits = dgrad_wgrad_pick(M, N, K)
    blocks = (M // 256) * (N // 256) + (K // 256) * (N // 256) * splits
    assert paired == 1 and bm == 256 and -(-blocks // 256) == rounds, (paired, bm, splits, blocks)
    DY, W, X = dev(dy, torch.bfloat16), dev(w, torch.bfloat16), dev(x, torch.bfloat16)
    dx = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    cs = torch.zeros((M // bm, N), dtype=torch.float32, device="cuda")
    dw = torch.full((splits, K, N), 7.0, dtype=torch.float32, device="cuda")
    L.rv_linear_dgrad_wgrad(DY.data_ptr(), K, W.data_ptr(), N, X.data_ptr(), N, M, N, K, dx.data_ptr(), N,
                            cs.data_ptr(), dw.data_ptr(), N, splits, 0, None, sp())
    # float64 on the GPU (a 12288 x 2048 x 512 contraction takes numpy tens of seconds)
    d64, w64, x64 = DY.double(), W.double(), X.double()
    ref_dx = (d64 @ w64) * (x64 > 0)
    ref_dw = d64.T @ x64
    assert float((dx.double() - ref_dx).abs().max()) <= 2 ** -7 * float(ref_dx.abs().max())
    assert torch.allclose(cs.double().sum(0), ref_dx.sum(0), rtol=1e-4, atol=1e-4 * float(ref_dx.sum(0).abs().max()))
    assert torch.allclose(dw.double().sum(0), ref_dw, rtol=1e-5, atol=1e-5 * float(ref_dw.abs().max()))


@pytest.mark.parametrize("M,N,K,ds,ws", [(512, 64, 256, 2, 4), (4096, 64, 2048, 4, 8), (256, 256, 512, 1, 2), (384, 128, 192, 1, 1)])
def test_linear_dgrad_wgrad_f32(L, M, N, K, ds, ws):
    """rv_linear_dgrad_wgrad_f32: dX = dY W as fp32 split slabs and dW = dY^T X slabs -- one dual launch when
    both GEMMs run on the same small tile (N = 64 / 128), two launches otherwise; M = batch, K = out, N = in."""
    rng = np.random.default_rng(43)
    dy, w, x = rand_bf16(rng, (M, K)), rand_bf16(rng, (K, N), 0.1), rand_bf16(rng, (M, N))
    DY, W, X = dev(dy, torch.bfloat16), dev(w, torch.bfloat16), dev(x, torch.bfloat16)
    dx = torch.full((ds, M, N), 3.0, dtype=torch.float32, device="cuda")
    dw = torch.full((ws, K, N), 7.0, dtype=torch.float32, device="cuda")
    L.rv_linear_dgrad_wgrad_f32(DY.data_ptr(), K, W.data_ptr(), N, X.data_ptr(), N, M, N, K, dx.data_ptr(), N, ds,
                                dw.data_ptr(), N, ws, sp())
    ref_dx = dy.astype(np.float64) @ w.astype(np.float64)
    ref_dw = dy.astype(np.float64).T @ x.astype(np.float64)
    np.testing.assert_allclose(dx.sum(0).cpu().numpy(), ref_dx, rtol=1e-5, atol=1e-5 * np.abs(ref_dx).max())
    np.testing.assert_allclose(dw.sum(0).cpu().numpy(), ref_dw, rtol=1e-5, atol=1e-5 * np.abs(ref_dw).max())


def test_heads_backward_dual_launch(L):
    """rv_linear_dgrad_wgrad where the 256x256 pairing does not apply (K = 2 Lp = 128, the heads): the
    dgrad takes the wgrad's 128x128 tile and both go out in one launch; column sums follow that tile."""
    from rawaudiovae_kelsey_amd._lib import dgrad_wgrad_pick
    M, N, K = 1024, 512, 128
    rng = np.random.default_rng(44)
    dy, w = rand_bf16(rng, (M, K)), rand_bf16(rng, (K, N), 0.1)
    x = O.bf16_round(np.maximum(rng.standard_normal((M, N)), 0).astype(np.float32))
    paired, bm, splits = dgrad_wgrad_pick(M, N, K)
    assert paired == 0 and bm == 128
    DY, W, X = dev(dy, torch.bfloat16), dev(w, torch.bfloat16), dev(x, torch.bfloat16)
    dx = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    cs = torch.zeros((M // bm, N), dtype=torch.float32, device="cuda")
    dw = torch.full((splits, K, N), 7.0, dtype=torch.float32, device="cuda")
    L.rv_linear_dgrad_wgrad(DY.data_ptr(), K, W.data_ptr(), N, X.data_ptr(), N, M, N, K, dx.data_ptr(), N,
                            cs.data_ptr(), dw.data_ptr(), N, splits, 0, None, sp())
    ref_dx = (dy.astype(np.float64) @ w.astype(np.float64)) * (x > 0)
    ref_dw = dy.astype(np.float64).T @ x.astype(np.float64)
    assert np.abs(dx.float().cpu().numpy() - ref_dx).max() <= 2 ** -7 * np.abs(ref_dx).max()
    np.testing.assert_allclose(cs.sum(0).cpu().numpy(), ref_dx.sum(0), rtol=1e-4, atol=1e-4 * np.abs(ref_dx.sum(0)).max())
    np.testing.assert_allclose(dw.sum(0).cpu().numpy(), ref_dw, rtol=1e-5, atol=1e-5 * np.abs(ref_dw).max())


def test_bf16_gradient_payload_kernels(L):
    """rv_grad_finalize(out_bf16) == bf16(rv_grad_finalize), and rv_adam_multi(grad_bf16) == rv_adam_multi fed the
    same (already bf16-valued) gradient."""
    from rawaudiovae_kelsey_amd._lib import ParamDesc
    rng = np.random.default_rng(45)
    rows, cols, splits = 37, 50, 3
    slabs = dev(rng.standard_normal((splits, rows, cols)).astype(np.float32))
    d = (ParamDesc * 1)(ParamDesc(8, rows, cols, slabs.data_ptr(), cols, rows * cols, splits, None, None, 0))
    n = 8 + rows * cols + 5
    g32 = torch.zeros(n, device="cuda")
    g16 = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    L.rv_grad_finalize(d, 1, g32.data_ptr(), 0, sp())
    L.rv_grad_finalize(d, 1, g16.data_ptr(), 1, sp())
    assert torch.equal(g16, g32.to(torch.bfloat16))
    step = torch.tensor([3], dtype=torch.int64, device="cuda")
    outs = []
    for use_bf16 in (False, True):
        p = dev(rng.standard_normal(n).astype(np.float32) * 0 + 0.5)
        m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        if use_bf16:
            L.rv_adam_multi(d, 1, p.data_ptr(), m.data_ptr(), v.data_ptr(), None, g16.data_ptr(), 1e-3, 0.5,
                                     step.data_ptr(), sp())
        else:
            flat = g16.float()
            df = (ParamDesc * 1)(ParamDesc(8, rows, cols, flat.data_ptr() + 4 * 8, cols, 0, 1, None, None, 0))
            L.rv_adam_multi(df, 1, p.data_ptr(), m.data_ptr(), v.data_ptr(), None, None, 1e-3, 0.5, step.data_ptr(), sp())
        torch.cuda.synchronize()
        outs.append((p.clone(), m.clone(), v.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_scale_by3_one_launch_any_alignment():
    """rv_scale_by3 (loss_function's backward: three saved gradients times the upstream scalar in one launch): exact
    products for aligned and unaligned pointers, ragged counts, a skipped tensor and all three skipped."""
    from rawaudiovae_kelsey_amd._lib import lib, ptr, stream_ptr
    g = torch.tensor([-1.75], device="cuda")
    base = torch.randn(3 * 70001 + 64, device="cuda")
    for off in (0, 1, 3):                       # element offsets: 16-byte aligned, and not
        a0 = base[off:off + 70001]
        a1 = base[70001 + off:70001 + off + 4099]
        a2 = base[2 * 70001 + off:2 * 70001 + off + 8]
        o0, o1, o2 = torch.zeros(70001 + 8, device="cuda")[off:off + 70001], torch.empty_like(a1), torch.empty_like(a2)
        lib().rv_scale_by3(ptr(a0), ptr(o0), a0.numel(), ptr(a1), ptr(o1), a1.numel(), ptr(a2), ptr(o2), a2.numel(), ptr(g),
                           stream_ptr())
        for a, o in ((a0, o0), (a1, o1), (a2, o2)):
            assert torch.equal(o, a * g)
    o1.zero_()
    lib().rv_scale_by3(None, None, 0, ptr(a1), ptr(o1), a1.numel(), None, None, 0, ptr(g), stream_ptr())
    assert torch.equal(o1, a1 * g)
    lib().rv_scale_by3(None, None, 0, None, None, 0, None, None, 0, ptr(g), stream_ptr())
    with pytest.raises(Exception):
        lib().rv_scale_by3(ptr(a0), None, 5, None, None, 0, None, None, 0, ptr(g), stream_ptr())
    torch.cuda.synchronize()


def test_latent_fwd_heads_only_equals_the_full_kernel(L):
    """rv_latent_fwd with w3 == NULL (heads + reparameterisation only: fc3 is generated inside the fc4 forward) writes the
    same mu | logvar, z, KL partials and eps as the full kernel, bit for bit."""
    rng = np.random.default_rng(4)
    for (B, Lt, H) in [(4096, 64, 2048), (200, 20, 512), (48, 64, 1024)]:
        Bp, Lp, Hp = -(-B // 16) * 16, 64, H
        h = np.zeros((Bp, Hp), np.float32); h[:B] = np.maximum(rand_bf16(rng, (B, H), 0.5), 0)
        wh = np.zeros((2 * Lp, Hp), np.float32)
        wh[:Lt] = rand_bf16(rng, (Lt, H), 0.05); wh[Lp:Lp + Lt] = rand_bf16(rng, (Lt, H), 0.05)
        bh = (rng.standard_normal(2 * Lp) * 0.1).astype(np.float32)
        w3 = rand_bf16(rng, (Hp, Lp), 0.2); b3 = (rng.standard_normal(Hp) * 0.1).astype(np.float32)
        hd, whd, w3d, bhd, b3d = dev(h, torch.bfloat16), dev(wh, torch.bfloat16), dev(w3, torch.bfloat16), dev(bh), dev(b3)
        ctr = torch.ones(1, dtype=torch.int64, device="cuda")
        res = []
        for heads_only in (False, True):
            mulv = torch.full((Bp, 2 * Lp), 5.0, device="cuda")
            zz = torch.full((Bp, Lp), 5.0, device="cuda", dtype=torch.bfloat16)
            kl = torch.zeros(Bp * Lp // 1024 + 1, device="cuda")
            e = torch.empty(B, Lt, device="cuda")
            h3 = torch.empty(Bp, Hp, device="cuda", dtype=torch.bfloat16)
            L.rv_latent_fwd(hd.data_ptr(), Hp, whd.data_ptr(), Hp, bhd.data_ptr(), None if heads_only else w3d.data_ptr(), Lp,
                            None if heads_only else b3d.data_ptr(), Bp, Hp, Lp, B, Lt, None, e.data_ptr(), 5, ctr.data_ptr(),
                            mulv.data_ptr(), zz.data_ptr(), kl.data_ptr(), None if heads_only else h3.data_ptr(), Hp, sp())
            res.append((mulv, zz, kl, e))
        torch.cuda.synchronize()
        for x0, x1 in zip(*res):
            assert torch.equal(x0, x1)
