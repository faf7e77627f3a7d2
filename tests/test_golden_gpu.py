"""HIP path against the reference's own golden vectors at the benchmark shape, and the Adam
state it leaves after 1 and 3 steps (GPU).

  * C2 (S=1024, H=2048, L=64, B=4096 -- BASELINE configs[1], the shape bench.py times, where the
    paired 256x256 ping-pong kernel, split-K and the XCD tile order are live): one
    FWD | BWD | FINALIZE step on the fixture seeds (frames 1234, eps 4321).  recon / mu / logvar and
    all ten gradients are compared with the L2 norms and the 16 sampled elements that
    tools/make_golden.py captured from the reference (`summary.json["cases"]["c2_f32"]`), and with the
    oracle run at the same bf16 rounding points (activations ~1 bf16 ulp, gradients 5e-3 rel-L2).
  * small shape: engine `param / exp_avg / exp_avg_sq` after 1 and 3 full steps against
    `small_f32.npz["after{1,3}/..."]` (the reference's torch.optim.Adam state, train.py:163,184-193)
    with relative bounds, and against the bf16-quantised oracle more tightly.

Tolerances vs the fp32 reference (bf16 MFMA inputs, fp32 accumulation):
loss 1e-4 rel; recon/mu/logvar 1e-2 abs per element and 2e-3 rel on the L2 norm; gradients 6e-2
rel-L2 (ReLU-mask flips from the bf16 rounding of h1/h3) -- for a fixture that holds only a norm and
samples this reads: norm within 3e-2, each sample within (what the oracle with and without the bf16 rounding
points says those operands do to that element) + 2e-2 of the tensor's rms + 2e-2 of its value, capped at 15e-2 rms.
"""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import GOLDEN  # noqa: E402
from oracle import vae_oracle as O  # noqa: E402
from oracle.inputs import PARAM_NAMES, make_eps, make_frames, make_params  # noqa: E402

KL, LR = 1e-4, 1e-4


def _engine(S, H, L, B, **kw):
    from rawaudiovae_kelsey_amd.engine import TrainEngine
    e = TrainEngine(S, H, L, B, kl_beta=KL, lr=LR, **kw)
    e.load_params(make_params(S, H, L, 0))
    return e


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


@pytest.fixture(scope="module", params=["c2_f32", "refini_f32"])
def c2_step(request):
    """One forward + backward on the HIP path and on the bf16-quantised oracle (same inputs), at C2 and at the
    reference's own configuration (`refini`: S=1024 H=2048 L=256 -- default.ini:3,18-19 / kelsey_iterable.ini:17-18 --,
    batch 4096 as kelsey_iterable.ini:26), where the latent-sized launches are the GEMM forms with fused
    reparameterisation epilogues instead of C2's row-local kernels."""
    from rawaudiovae_kelsey_amd import engine as E
    with open(os.path.join(GOLDEN, "summary.json")) as f:
        cs = json.load(f)["cases"][request.param]
    S, H, L, B = cs["shape"]
    e = _engine(S, H, L, B)
    x, eps = make_frames(B, S, 1234), make_eps(B, L, 4321)
    recon = torch.zeros(B, S, device="cuda")
    e.step(torch.from_numpy(x).cuda(), torch.from_numpy(eps).cuda(), recon,
           phases=E.PHASE_FWD | E.PHASE_BWD_A | E.PHASE_BWD_B | E.PHASE_FINALIZE_A | E.PHASE_FINALIZE_B)
    torch.cuda.synchronize()
    mu, lv = e.outputs()
    got = {"recon": recon.cpu().numpy(), "mu": mu.cpu().numpy(), "logvar": lv.cpu().numpy()}
    got.update({"grad/" + k: v.cpu().numpy() for k, v in e.grad_views().items()})
    p = O.cast_params(make_params(S, H, L, 0), np.float32)
    c = O.forward(p, x, eps, quant="bf16")
    g = O.backward(p, c, KL, quant="bf16")
    orc = {"recon": c["recon"], "mu": c["mu"], "logvar": c["logvar"]}
    orc.update({"grad/" + k: g[k] for k in PARAM_NAMES})
    # the oracle WITHOUT the rounding points (the fp32 restatement tests/test_oracle_golden.py pins to the reference at
    # this very shape): q - f at an element is what bf16 operands do to THAT element
    cf = O.forward(p, x, eps)
    gf = O.backward(p, cf, KL)
    orc.update({"f32grad/" + k: gf[k] for k in PARAM_NAMES})
    oloss = O.loss_function(c["recon"].astype(np.float64), x.astype(np.float64), c["mu"].astype(np.float64),
                            c["logvar"].astype(np.float64), KL)[0]
    return cs, got, orc, e.last_loss()[0], oloss


def test_c2_loss_vs_reference_and_oracle(c2_step):
    cs, _, _, loss, oloss = c2_step
    assert abs(loss - cs["loss0"]) <= 1e-4 * cs["loss0"], (loss, cs["loss0"])
    assert abs(loss - oloss) <= 2e-5 * abs(oloss), (loss, oloss)


@pytest.mark.parametrize("name", ["recon", "mu", "logvar"])
def test_c2_activations_vs_reference_golden(c2_step, name):
    cs, got, orc, _, _ = c2_step
    info = cs["tensors"][name]
    flat = got[name].reshape(-1).astype(np.float64)
    assert abs(np.linalg.norm(flat) - info["l2"]) <= 2e-3 * info["l2"]
    np.testing.assert_allclose(flat[info["idx"]], info["val"], rtol=0, atol=1e-2)
    # same rounding points: ~1 bf16 ulp of the tensor scale.  At this size the fp32 summation order of a
    # 1024- or 2048-deep contraction differs between MFMA tiles and numpy, which now and then flips the bf16
    # rounding of one h1/h3 element (one bf16 ulp = 0.4 %); an output fed by it moves by up to a few 1e-3.
    # Hence two bounds: nearly every element within 1 ulp-of-scale, every element within 5e-3.
    scale = max(1.0, float(np.abs(orc[name]).max()))
    err = np.abs(got[name].astype(np.float64) - orc[name])
    tight = 5e-4 if name == "recon" else 2e-4 * scale
    assert float((err > tight).mean()) < 5e-3, float((err > tight).mean())
    assert float(err.max()) < 5e-3 * scale, float(err.max())


@pytest.mark.parametrize("name", PARAM_NAMES)
def test_c2_gradients_vs_reference_golden(c2_step, name):
    cs, got, orc, _, _ = c2_step
    info = cs["tensors"]["grad/" + name]
    g = got["grad/" + name]
    flat = g.reshape(-1).astype(np.float64)
    rms = info["l2"] / np.sqrt(flat.size)
    assert abs(np.linalg.norm(flat) - info["l2"]) <= 3e-2 * info["l2"], (np.linalg.norm(flat), info["l2"])
    err = np.abs(flat[info["idx"]] - np.array(info["val"]))
    # element by element: the distance to the fp32 reference is what the oracle says bf16 operands do to that very element
    # (|q - f| at the sampled index, rounding points on minus rounding points off) plus 2 % -- and never more than 15 % of the
    # tensor's rms.  (Round 5 held every sample to a flat 6 % of the rms; at the reference's own latent width one of the 16
    # fc1.weight samples sits at 7.5 % with the rounding effect the oracle predicts for it, on the old and the new kernels alike.)
    qf = np.abs(orc["grad/" + name].reshape(-1).astype(np.float64)[info["idx"]] -
                orc["f32grad/" + name].reshape(-1).astype(np.float64)[info["idx"]])
    assert np.all(err <= qf + 2e-2 * rms + 2e-2 * np.abs(info["val"])), ((err - qf).max(), rms)
    assert np.all(err <= 15e-2 * rms + 6e-2 * np.abs(info["val"])), (err.max(), rms)
    # against the oracle with the HIP path's bf16 rounding points: every element of the gradient
    assert _rel_l2(g, orc["grad/" + name]) < 5e-3


def test_adam_state_vs_reference_golden():
    """param / exp_avg / exp_avg_sq after 1 and 3 steps against the reference's optimizer state
    (small_f32.npz) and against the bf16-quantised oracle.  The bound on the PARAMETERS against the fp32 reference is
    wide on purpose (2 * lr * n and rel-L2 0.15 on the update): Adam's first steps are ~sign(g) * lr, so an element
    whose bf16-path gradient has the other sign moves the other way by a full lr.  The tight evidence for the optimizer
    arithmetic is elsewhere: tests/test_strict_fp32_gpu.py (fp32 operands: update within 1e-5 of the reference's) and
    tests/test_kernels_gpu.py::test_adam_multi_and_finalize (the kernel against numpy on identical gradients)."""
    g = np.load(os.path.join(GOLDEN, "small_f32.npz"))
    S, H, L, B = (int(v) for v in g["shape"])
    e = _engine(S, H, L, B)
    p0 = make_params(S, H, L, 0)
    pq = O.cast_params(make_params(S, H, L, 0), np.float32)
    stq = O.adam_init(pq)
    for i in range(3):
        x, eps = make_frames(B, S, 1234 + i), make_eps(B, L, 4321 + i)
        e.step(torch.from_numpy(x).cuda(), torch.from_numpy(eps).cuda())
        O.train_step(pq, stq, x, eps, KL, LR, quant="bf16")
        n = i + 1
        if n not in (1, 3):
            continue
        torch.cuda.synchronize()
        for k in PARAM_NAMES:
            m = e.view(e.exp_avg, k).cpu().numpy()
            v = e.view(e.exp_avg_sq, k).cpu().numpy()
            w = e.view(e.param, k).cpu().numpy()
            rm, rv_, rw = (g["after%d/%s/%s" % (n, t, k)] for t in ("exp_avg", "exp_avg_sq", "param"))
            # vs the fp32 reference: the moments inherit the gradient tolerance (6e-2 rel-L2; squared: 1.2e-1)
            assert _rel_l2(m, rm) < 6e-2, (n, k, _rel_l2(m, rm))
            assert _rel_l2(v, rv_) < 1.2e-1, (n, k, _rel_l2(v, rv_))
            # the update m/(sqrt(v)+eps) is ~sign(g) on the first step: elements move by ~lr each, and
            # only near-zero gradients (sign flips under bf16 rounding) differ, by at most 2 lr per step
            upd, rupd = w.astype(np.float64) - p0[k], rw.astype(np.float64) - p0[k]
            assert np.abs(upd - rupd).max() <= 2.0 * LR * n + 1e-9, (n, k)
            assert _rel_l2(upd, rupd) < 0.15, (n, k, _rel_l2(upd, rupd))
            assert abs(np.linalg.norm(upd) - np.linalg.norm(rupd)) <= 2e-2 * np.linalg.norm(rupd)
            if n == 1:
                # same rounding points: tight.  (From step 2 on the weights themselves differ by sign-flip
                # updates, so the two runs see different bf16 weights and only the bounds above apply.)
                assert _rel_l2(m, stq["exp_avg"][k]) < 5e-3, (n, k)
                assert _rel_l2(v, stq["exp_avg_sq"][k]) < 1e-2, (n, k)
                assert _rel_l2(w.astype(np.float64) - p0[k], pq[k].astype(np.float64) - p0[k]) < 5e-2, (n, k)
    assert int(e.step_counter.item()) == 3


def test_device_framing_vs_reference_dataset_fixtures():
    """rv_gather_frames / data.DeviceAudio / DeviceEvalAudio bit-exact against frame matrices produced by
    running the reference's AudioDataset / TestDataset (rawvae/dataset.py:86-160; fixtures from
    tools/make_golden.py:dataset_case), incl. the zero-padded tail and a shuffled index."""
    import zlib
    from rawaudiovae_kelsey_amd import data as D
    fx = np.load(os.path.join(GOLDEN, "dataset_frames.npz"))
    ramp = np.arange(1000, dtype=np.float32)
    w2 = np.random.default_rng(int(fx["rand_wave_seed"])).uniform(-1, 1, 5000).astype(np.float32)
    for wave, S, hop, key in ((ramp, 256, 64, "ramp"), (w2, 512, 128, "rand")):
        ref = fx[key + "_hop_frames"]
        d = D.DeviceAudio(wave, S, hop)
        assert len(d) == ref.shape[0]
        np.testing.assert_array_equal(d.frames(0, len(d)).cpu().numpy(), ref)
        perm = torch.randperm(len(d), generator=torch.Generator().manual_seed(3))
        np.testing.assert_array_equal(d.gather(perm.cuda()).cpu().numpy(), ref[perm.numpy()])
        ev = D.DeviceEvalAudio(wave, S)
        np.testing.assert_array_equal(ev.frames(0, len(ev)).cpu().numpy(), fx[key + "_eval_frames"])
    with open(os.path.join(GOLDEN, "summary.json")) as f:
        ds = json.load(f)["dataset"]
    wave = np.random.default_rng(ds["wave_seed"]).uniform(-1, 1, ds["n_samples"]).astype(np.float32)
    d = D.DeviceAudio(wave, ds["segment_length"], ds["hop"])
    assert (len(d), d.padded) == (ds["len"], ds["padded"])
    idx = torch.tensor(ds["frame_idx"], dtype=torch.int64, device="cuda")
    got = d.gather(idx).cpu().numpy()
    assert [int(zlib.crc32(r.tobytes())) for r in got] == ds["frame_crc32"]
    batches = list(d.batches(ds["batch"], shuffle=False))
    assert batches[-1].shape[0] == ds["last_batch"]
    assert int(zlib.crc32(batches[-1].cpu().numpy().tobytes())) == ds["last_batch_crc32"]
    with pytest.raises(ValueError) as ei:
        D.DeviceAudio(wave, ds["bad_segment_length"], ds["hop"])
    assert str(ei.value) == ds["bad_segment_raises"]


@pytest.mark.parametrize("shape", [(512, 384, 12, 128), (256, 1024, 64, 64), (500, 256, 8, 100), (192, 256, 8, 64)])
@pytest.mark.parametrize("fp8", [False, True])
def test_step_on_resident_waveform_equals_step_on_gathered_frames(fp8, shape):
    """N1 as SURVEY 8f specifies it: `step_frames` (fc1's GEMM tile loader reading frame i at i * hop of the resident
    bf16 waveform -- no cast kernel --, fc4's loss epilogue reading its fp32 target there) gives bit for bit what
    gather -> `step` gives, for a shuffled index with the zero-padded tail frames in it, a ragged batch, and
    consecutive frames.  Shapes: hop a multiple of 8 (the in-place path; with fp8 the cast kernel that also quantises),
    a hidden width that gets the 256 x 128 tile, hop = 100 with S = 500 (16-byte pieces impossible: cast-kernel
    fallback, padded columns), and S = 192 with hop 64 (an aligned hop but S < Sp = 256: the padded columns must be
    zeros, not the 64 samples behind the frame, so the plan takes the cast-kernel route there too)."""
    from rawaudiovae_kelsey_amd import data as D
    from rawaudiovae_kelsey_amd.engine import TrainEngine
    S, H, L, hop = shape
    wave = np.random.default_rng(5).uniform(-1, 1, 40000).astype(np.float32)
    d = D.DeviceAudio(wave, S, hop)
    n = len(d)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1)).cuda()
    for B, idx in ((200, perm[:200].contiguous()), (n - 200, perm[200:].contiguous())):
        a = TrainEngine(S, H, L, B, kl_beta=KL, lr=LR, seed=2, fp8=fp8)
        b = TrainEngine(S, H, L, B, kl_beta=KL, lr=LR, seed=2, fp8=fp8)
        for e in (a, b):
            e.load_params(make_params(S, H, L, 0))
        eps = torch.from_numpy(make_eps(B, L, 9)).cuda()
        ra, rb = torch.zeros(B, S, device="cuda"), torch.zeros(B, S, device="cuda")
        for _ in range(2):
            a.step(d.gather(idx), eps, ra)
            b.step_frames(d, idx, eps=eps, recon_out=rb)
        torch.cuda.synchronize()
        assert torch.equal(a.param, b.param) and torch.equal(ra, rb) and a.losses(2) == b.losses(2)
    # consecutive frames, incl. the last (zero-padded) ones
    B = 64
    a = TrainEngine(S, H, L, B, kl_beta=KL, lr=LR, seed=2, fp8=fp8)
    b = TrainEngine(S, H, L, B, kl_beta=KL, lr=LR, seed=2, fp8=fp8)
    for e in (a, b):
        e.load_params(make_params(S, H, L, 0))
    a.step(d.frames(n - B, B))
    b.step_frames(d, first_frame=n - B)
    torch.cuda.synchronize()
    assert torch.equal(a.param, b.param) and a.losses(1) == b.losses(1)
