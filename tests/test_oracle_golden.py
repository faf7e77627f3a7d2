"""Pin the CPU oracle against outputs of the reference itself
(tests/golden/*, produced by tools/make_golden.py from /root/reference)."""
import json
import os

import numpy as np
import pytest

from oracle import vae_oracle as O
from oracle.inputs import PARAM_NAMES, make_eps, make_frames, make_params

from conftest import GOLDEN


def _load(tag):
    return np.load(os.path.join(GOLDEN, "small_%s.npz" % tag))


def _run_small(dtype, n_steps):
    S, H, L, B = 64, 96, 8, 16
    params = O.cast_params(make_params(S, H, L, 0), dtype)
    state = O.adam_init(params)
    out = []
    for i in range(n_steps):
        x = make_frames(B, S, 1234 + i).astype(dtype)
        eps = make_eps(B, L, 4321 + i).astype(dtype)
        loss, c, grads = O.train_step(params, state, x, eps, 1e-4, 1e-4)
        out.append((loss, c, grads, {k: v.copy() for k, v in params.items()},
                    {k: v.copy() for k, v in state["exp_avg"].items()},
                    {k: v.copy() for k, v in state["exp_avg_sq"].items()}))
    return out


@pytest.mark.parametrize("tag,dtype,rtol", [("f64", np.float64, 1e-11), ("f32", np.float32, 2e-5)])
def test_small_forward_loss_grads(tag, dtype, rtol):
    g = _load(tag)
    loss, c, grads, *_ = _run_small(dtype, 1)[0]
    assert abs(loss - float(g["loss"])) <= rtol * abs(float(g["loss"]))
    for k in ("recon", "mu", "logvar"):
        np.testing.assert_allclose(c[k], g[k], rtol=rtol, atol=rtol * np.abs(g[k]).max())
    for k in PARAM_NAMES:
        ref = g["grad/" + k]
        np.testing.assert_allclose(grads[k], ref, rtol=rtol * 10, atol=rtol * np.abs(ref).max())


@pytest.mark.parametrize("tag,dtype,rtol", [("f64", np.float64, 1e-9), ("f32", np.float32, 1e-4)])
def test_small_adam_and_trajectory(tag, dtype, rtol):
    g = _load(tag)
    runs = _run_small(dtype, 20)
    traj = np.array([r[0] for r in runs])
    np.testing.assert_allclose(traj, g["traj"], rtol=rtol)
    for n in (1, 3):
        _, _, _, params, m, v = runs[n - 1]
        for k in PARAM_NAMES:
            np.testing.assert_allclose(params[k], g["after%d/param/%s" % (n, k)], rtol=rtol,
                                       atol=rtol * 1e-2)
            np.testing.assert_allclose(m[k], g["after%d/exp_avg/%s" % (n, k)], rtol=rtol * 10,
                                       atol=rtol * np.abs(m[k]).max())
            np.testing.assert_allclose(v[k], g["after%d/exp_avg_sq/%s" % (n, k)], rtol=rtol * 10,
                                       atol=rtol * np.abs(v[k]).max())


@pytest.mark.parametrize("case", ["smoke_f64", "smoke_f32", "c2_f64", "c2_f32", "refini_f64", "refini_f32"])
def test_summary_smoke(case):
    """Smoke shape, the benchmark shape C2 (1024, 2048, 64, 4096) and the reference's own model (`refini`: 1024, 2048,
    256 -- /root/reference/default.ini:3,18-19, kelsey_iterable.ini:17-18 -- at kelsey_iterable.ini:26's batch 4096): the
    oracle is pinned to the reference at the sizes bench.py runs, not only at toy sizes."""
    with open(os.path.join(GOLDEN, "summary.json")) as f:
        summ = json.load(f)
    cs = summ["cases"][case]
    S, H, L, B = cs["shape"]
    dtype = np.float64 if case.endswith("f64") else np.float32
    rtol = 1e-10 if dtype == np.float64 else 5e-5
    params = O.cast_params(make_params(S, H, L, 0), dtype)
    state = O.adam_init(params)
    for i in range(3 if case.startswith("smoke") else 2):
        x = make_frames(B, S, 1234 + i).astype(dtype)
        eps = make_eps(B, L, 4321 + i).astype(dtype)
        loss, c, grads = O.train_step(params, state, x, eps, summ["kl_beta"], summ["lr"])
        assert abs(loss - cs["traj"][i]) <= rtol * abs(cs["traj"][i])
        if i == 0:
            tensors = {"recon": c["recon"], "mu": c["mu"], "logvar": c["logvar"]}
            tensors.update({"grad/" + k: grads[k] for k in PARAM_NAMES})
            for k, info in cs["tensors"].items():
                flat = tensors[k].reshape(-1).astype(np.float64)
                assert abs(np.sqrt((flat ** 2).sum()) - info["l2"]) <= 10 * rtol * info["l2"]
                np.testing.assert_allclose(flat[info["idx"]], info["val"], rtol=100 * rtol,
                                           atol=10 * rtol * np.abs(flat).max())


def test_torch_port_matches_golden_small():
    """The torch fp32 port that bench.py times as cpu_baseline is the same math."""
    torch = pytest.importorskip("torch")
    from oracle.torch_port import PortVAE, port_loss
    g = _load("f32")
    S, H, L, B = 64, 96, 8, 16
    m = PortVAE(S, H, L).load_numpy(make_params(S, H, L, 0))
    x = torch.from_numpy(make_frames(B, S, 1234))
    eps = torch.from_numpy(make_eps(B, L, 4321))
    recon, mu, logvar = m(x, eps)
    loss = port_loss(recon, x, mu, logvar, 1e-4, S)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    np.testing.assert_allclose(recon.detach().numpy(), g["recon"], atol=1e-6)
    sd = dict(m.named_parameters())
    for k in PARAM_NAMES:
        np.testing.assert_allclose(sd[k].grad.numpy(), g["grad/" + k], rtol=1e-4,
                                   atol=1e-6 * np.abs(g["grad/" + k]).max() + 1e-12)


def test_bf16_round_is_rne():
    a = np.array([1.0, 1.00390625, 1.005859375, -3.1415927, 1e-30, 65504.0], dtype=np.float32)
    r = O.bf16_round(a)
    torch = pytest.importorskip("torch")
    t = torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    np.testing.assert_array_equal(r, t)


def test_dataset_known_answers():
    """AudioDataset / TestDataset framing semantics against fixtures produced by RUNNING the reference's
    classes (rawvae/dataset.py:86-160; tools/make_golden.py:dataset_case): length / padding / ragged last
    batch / ValueError text for 30 s @ 44.1 kHz, frame checksums, and complete frame matrices."""
    import zlib
    with open(os.path.join(GOLDEN, "summary.json")) as f:
        d = json.load(f)["dataset"]
    n, padded = O.frame_count(d["n_samples"], d["segment_length"], d["hop"])
    assert (n, padded) == (d["len"], d["padded"])
    assert n % d["batch"] == d["last_batch"]
    with pytest.raises(ValueError) as ei:
        O.frame_count(d["n_samples"], d["bad_segment_length"], d["hop"])
    assert str(ei.value) == d["bad_segment_raises"]
    wave = np.random.default_rng(d["wave_seed"]).uniform(-1, 1, d["n_samples"]).astype(np.float32)
    fr = O.hop_frames(wave, d["segment_length"], d["hop"], d["frame_idx"])
    assert [int(zlib.crc32(r.tobytes())) for r in fr] == d["frame_crc32"]
    last = O.hop_frames(wave, d["segment_length"], d["hop"], np.arange(n - d["last_batch"], n))
    assert int(zlib.crc32(np.ascontiguousarray(last).tobytes())) == d["last_batch_crc32"]
    ev = O.eval_frames(wave, d["segment_length"])
    assert ev.shape[0] == d["test_len"] and ev.size == d["test_padded"]
    assert int(zlib.crc32(np.ascontiguousarray(ev[-1]).tobytes())) == d["test_last_crc32"]
    fx = np.load(os.path.join(GOLDEN, "dataset_frames.npz"))
    ramp = np.arange(1000, dtype=np.float32)
    np.testing.assert_array_equal(O.hop_frames(ramp, 256, 64), fx["ramp_hop_frames"])
    np.testing.assert_array_equal(O.eval_frames(ramp, 256), fx["ramp_eval_frames"])
    w2 = np.random.default_rng(int(fx["rand_wave_seed"])).uniform(-1, 1, 5000).astype(np.float32)
    np.testing.assert_array_equal(O.hop_frames(w2, 512, 128), fx["rand_hop_frames"])
    np.testing.assert_array_equal(O.eval_frames(w2, 512), fx["rand_eval_frames"])
