"""fp8 (e4m3) forward path for fc1 / fc4 (BASELINE configs[4]; a build extension, SURVEY D4 -- the reference
has no reduced-precision mode, so the contract is the build's own oracle with the fp8 rounding points,
`oracle/vae_oracle.py` quant="fp8", exactly as quant="bf16" is for the bf16 path).

Stated tolerances:
  * vs the fp8-quantised oracle (same rounding points, fp32 accumulation): loss 2e-5 rel.  Per element the
    comparison is statistical: a last-bit difference in an fp32 sum (MFMA vs numpy order) that crosses an e4m3
    rounding boundary moves that h3 element by a whole fp8 step (6-12 %) and with it every output of its row,
    so recon is held to 3e-2 abs everywhere with 90 % of the elements within 1e-3, gradients to 3e-2 rel-L2;
  * vs the reference's fp32 golden vectors: loss within 2e-3 rel at the smoke and benchmark shapes (e4m3 carries
    3 mantissa bits: ~3 % rms per element, averaged over 1024- / 2048-deep contractions), 20-step trajectory 5e-3.
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


def _engine(S, H, L, B, fp8=True, **kw):
    from rawaudiovae_kelsey_amd.engine import TrainEngine
    e = TrainEngine(S, H, L, B, kl_beta=KL, lr=LR, fp8=fp8, **kw)
    e.load_params(make_params(S, H, L, 0))
    return e


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def test_fp8_cast_matches_e4m3_rounding():
    """rv_cast_pad_fp8 == numpy e4m3 rounding == torch.float8_e4m3fn, incl. saturation and subnormals."""
    from rawaudiovae_kelsey_amd._lib import lib, stream_ptr
    rng = np.random.default_rng(3)
    a = np.concatenate([rng.normal(0, 60, 4000), rng.uniform(-0.05, 0.05, 4000),
                        [0, 448, 500, -500, 2 ** -9, 2 ** -10, 0.0156, 464, 465, -447.9] + [0.0] * 22]).astype(np.float32)
    a = a.reshape(-1, 16)
    rows, cols = a.shape
    src = torch.from_numpy(a).cuda()
    dst = torch.zeros((rows, cols), dtype=torch.uint8, device="cuda")
    scale = torch.tensor([1.0], device="cuda")
    lib().rv_cast_pad_fp8(src.data_ptr(), rows, cols, cols, dst.data_ptr(), rows, cols, cols, scale.data_ptr(), stream_ptr())
    got = dst.view(torch.float8_e4m3fn).float().cpu().numpy()
    np.testing.assert_array_equal(got, O.fp8_e4m3_round(a))
    np.testing.assert_array_equal(got, torch.from_numpy(np.clip(a, -448, 448)).to(torch.float8_e4m3fn).float().numpy())


@pytest.mark.parametrize("mode", ["full", "fwd"])
@pytest.mark.parametrize("shape", [(512, 2048, 8, 32), (256, 384, 100, 130), (1024, 2048, 64, 4096), (256, 512, 16, 1024)])
def test_fp8_step_vs_fp8_oracle(shape, mode):
    """One step against the oracle with the same rounding points.  mode "full": forward of fc1 / fc4 AND the backward of
    fc4 on e4m3 operands (where the paired 256 x 256 launch applies: C2; elsewhere the backward stays bf16 and the
    oracle follows); "fwd": the forward only."""
    from rawaudiovae_kelsey_amd import engine as E
    S, H, L, B = shape
    e = _engine(S, H, L, B, fp8=mode)
    e.set_fp8_scales(h3=32.0, freeze_h3=True)
    st = e.fp8_state()
    scales = {"x": st[0], "w1": st[1], "w4": st[2], "h3": 32.0}
    Bp_, Sp_, Hp_, _ = e.padded()
    from rawaudiovae_kelsey_amd._lib import dgrad_wgrad_pick
    paired, _, sp4 = dgrad_wgrad_pick(Bp_, Hp_, Sp_)
    # (the fp8 pair is the 256 x 256 paired launch: it exists where that pairing fills the chip -- C2 -- with an even number
    # of 128-deep K tiles per block; smaller shapes keep the bf16 backward and the oracle follows)
    f8_bwd = bool(mode == "full" and paired and (Sp_ // 128) % 2 == 0 and Bp_ % (128 * sp4) == 0 and (Bp_ // 128 // sp4) % 2 == 0)
    assert f8_bwd == (shape == (1024, 2048, 64, 4096) and mode == "full")
    assert abs(st[12] - 56.0 * B * S) <= 1e-6 * st[12]
    bwd_scales = dict(scales, dp4=st[12]) if f8_bwd else None
    x, eps = make_frames(B, S, 1234), make_eps(B, L, 4321)
    recon = torch.zeros(B, S, device="cuda")
    e.step(torch.from_numpy(x).cuda(), torch.from_numpy(eps).cuda(), recon,
           phases=E.PHASE_FWD | E.PHASE_BWD_A | E.PHASE_BWD_B | E.PHASE_FINALIZE_A | E.PHASE_FINALIZE_B)
    torch.cuda.synchronize()
    p = O.cast_params(make_params(S, H, L, 0), np.float32)
    c = O.forward(p, x, eps, quant="fp8", fp8_scales=scales)
    loss = O.loss_function(c["recon"].astype(np.float64), x.astype(np.float64), c["mu"].astype(np.float64),
                           c["logvar"].astype(np.float64), KL)[0]
    g = O.backward(p, c, KL, quant="fp8" if f8_bwd else "bf16", fp8_scales=bwd_scales)
    got = e.last_loss()[0]
    assert abs(got - loss) <= 2e-5 * abs(loss), (got, loss)
    err = np.abs(recon.cpu().numpy().astype(np.float64) - c["recon"])
    assert float(err.max()) < 3e-2 and float((err > 1e-3).mean()) < 0.1, (err.max(), (err > 1e-3).mean())
    gv = e.grad_views()
    for k in PARAM_NAMES:
        assert _rel_l2(gv[k].cpu().numpy(), g[k]) < 3e-2, (k, _rel_l2(gv[k].cpu().numpy(), g[k]))
    if f8_bwd:
        # the fp8 image of dP4 itself, and what the fp8 backward costs against the bf16 one: e4m3 carries 3 mantissa bits
        # (~2.5 % rms per element of dP4, averaged over 1024- / 4096-deep contractions)
        dp4q = e.buffer("dP4q", torch.uint8, (Bp_, Sp_)).view(torch.float8_e4m3fn).float().cpu().numpy()[:B, :S]
        rec = c["recon"].astype(np.float64)
        want = O.fp8_e4m3_round(((2.0 / (B * S)) * (rec - x) * (1.0 - rec * rec) * st[12]).astype(np.float32))
        # (recon itself carries the fp8 forward's noise against the oracle's -- 1e-3 typical --, so a few per cent of the
        # elements sit on the other side of an e4m3 rounding boundary; as a whole the image agrees)
        assert float((dp4q != want).mean()) < 0.15 and _rel_l2(dp4q, want) < 2e-2, ((dp4q != want).mean(), _rel_l2(dp4q, want))
        gb = O.backward(p, c, KL, quant="bf16")
        for k in ("fc4.weight", "fc3.weight", "fc1.weight", "fc21.weight"):
            assert _rel_l2(gv[k].cpu().numpy(), gb[k]) < 6e-2, (k, _rel_l2(gv[k].cpu().numpy(), gb[k]))
    # the fp8 operand images themselves: W1q is fp8(W1 * s_w1) zero-padded
    Bp, Sp, Hp, Lp = e.padded()
    w1q = e.buffer("W1q", torch.uint8, (Hp, Sp)).view(torch.float8_e4m3fn).float().cpu().numpy()
    np.testing.assert_array_equal(w1q[:H, :S], O.fp8_e4m3_round(p["fc1.weight"] * np.float32(st[1])))
    assert not w1q[H:].any() and not w1q[:, S:].any()


@pytest.mark.parametrize("case", ["smoke_f32", "c2_f32"])
def test_fp8_trajectory_vs_reference_golden(case):
    """20 full steps (delayed h3 scaling live, Adam rewriting the fp8 weight shadows) against the reference's
    fp32 loss trajectory."""
    with open(os.path.join(GOLDEN, "summary.json")) as f:
        cs = json.load(f)["cases"][case]
    S, H, L, B = cs["shape"]
    e = _engine(S, H, L, B)
    for i in range(20):
        e.step(torch.from_numpy(make_frames(B, S, 1234 + i)).cuda(), torch.from_numpy(make_eps(B, L, 4321 + i)).cuda())
    got, ref = np.array(e.losses(20)), np.array(cs["traj"])
    rel = np.abs(got - ref) / ref
    assert rel[0] <= 2e-3 and rel.max() <= 5e-3, rel
    st = e.fp8_state()
    assert st[3] != 16.0 and 1.0 < st[3] < 4096.0     # the activation scale has been latched from a measured max
    # Adam kept the fp8 shadow of fc4.weight equal to fp8(W4 * s_w4)
    Bp, Sp, Hp, Lp = e.padded()
    w4q = e.buffer("W4q", torch.uint8, (Sp, Hp)).view(torch.float8_e4m3fn).float().cpu().numpy()
    w4 = e.view(e.param, "fc4.weight").cpu().numpy()
    np.testing.assert_array_equal(w4q[:S, :H], O.fp8_e4m3_round(w4 * np.float32(st[2])))
    # ... with a scale that follows the weights: behind the optimizer a small kernel measures max|q| / scale of the two
    # fp8 shadows into slots; the next step's first kernel reduces them ([8], [9]) and moves the weight scales to
    # 224 / that maximum AFTER latching this step's dequantisation factors
    w1 = e.view(e.param, "fc1.weight").cpu().numpy()
    slots = e.buffer("fp8_state", torch.float32, (-1,))[32:].view(2, 1024).max(dim=1).values.tolist()
    assert abs(slots[1] * st[2] / float(np.abs(w4q).max()) - 1.0) < 1e-6                   # of the LAST update's shadow
    assert abs(slots[1] / float(np.abs(w4).max()) - 1.0) < 0.07 and abs(slots[0] / float(np.abs(w1).max()) - 1.0) < 0.07
    assert 0.9 < st[9] / slots[1] < 1.1 and 0.9 < st[8] / slots[0] < 1.1                  # of the one before it
    assert abs(st[2] * np.abs(w4).max() / 224.0 - 1.0) < 0.1 and np.abs(w4q).max() <= 448.0
    # a stale scale (weights that "grew" 8x since it was set) is gone after one step
    e.set_fp8_scales(w4=st[2] * 8.0)
    e.step(torch.from_numpy(make_frames(B, S, 99)).cuda(), torch.from_numpy(make_eps(B, L, 98)).cuda())
    st2 = e.fp8_state()
    assert abs(st2[2] * slots[1] / 224.0 - 1.0) < 1e-6
    w4q = e.buffer("W4q", torch.uint8, (Sp, Hp)).view(torch.float8_e4m3fn).float().cpu().numpy()
    w4 = e.view(e.param, "fc4.weight").cpu().numpy()
    np.testing.assert_array_equal(w4q[:S, :H], O.fp8_e4m3_round(w4 * np.float32(st2[2])))
    assert 200.0 < np.abs(w4q).max() <= 448.0


def test_fp8_full_step_weight_gradient_of_fc1_on_fp8_operands():
    """The full local step of the fp8 weight path at C2 also runs fc1's weight gradient on e4m3 operands: dP1 leaves the
    heads' backward as fp8 (no bf16 copy, nor one of the frames), the GEMM beside the optimizer riders multiplies it with
    the frames' fp8 image.  One step from zero moments leaves exp_avg = 0.1 g: all ten gradients against the oracle with
    the same rounding points (3e-2 rel-L2, as the phase-split test), fc1.weight's also against the bf16 backward (what
    the two fp8 operands cost), the image of dP1 itself, and the delayed scale: measured in step 1, used in step 2."""
    S, H, L, B = 1024, 2048, 64, 4096
    e = _engine(S, H, L, B, fp8="full")
    s_dp1 = 56.0 * B * S / 4.0
    e.set_fp8_scales(h3=32.0, freeze_h3=True, dp1=s_dp1)
    st = e.fp8_state()
    scales = {"x": st[0], "w1": st[1], "w4": st[2], "h3": 32.0, "dp4": st[12], "dp1": s_dp1}
    x, eps = make_frames(B, S, 1234), make_eps(B, L, 4321)
    e.step(torch.from_numpy(x).cuda(), torch.from_numpy(eps).cuda())
    torch.cuda.synchronize()
    p = O.cast_params(make_params(S, H, L, 0), np.float32)
    c = O.forward(p, x, eps, quant="fp8", fp8_scales=scales)
    g = O.backward(p, c, KL, quant="fp8", fp8_scales=scales)
    gb = O.backward(p, c, KL, quant="bf16")
    for k in PARAM_NAMES:
        got = e.view(e.exp_avg, k).cpu().numpy().astype(np.float64) / 0.1
        assert _rel_l2(got, g[k]) < 3e-2, (k, _rel_l2(got, g[k]))
    got = e.view(e.exp_avg, "fc1.weight").cpu().numpy().astype(np.float64) / 0.1
    assert _rel_l2(got, gb["fc1.weight"]) < 6e-2, _rel_l2(got, gb["fc1.weight"])
    # the image of dP1: the oracle's fp32 dP1 through the same scale (elements on the other side of an e4m3 rounding
    # boundary because dmu / dlv carry the fp8 forward's noise: a few per cent)
    Bp, Sp, Hp, Lp = e.padded()
    dp1q = e.buffer("dP1q", torch.uint8, (Bp, Hp)).view(torch.float8_e4m3fn).float().cpu().numpy()[:B, :H]
    q = lambda a: O.bf16_round(a)  # noqa: E731
    W21, W22 = q(p["fc21.weight"]), q(p["fc22.weight"])
    n_k = B * L
    dP4 = O._q8((2.0 / (B * S)) * (c["recon"] - x) * (1.0 - c["recon"] * c["recon"]), st[12])
    dP3 = q((dP4 @ O._q8(p["fc4.weight"], st[2])) * (c["h3"] > 0))
    dz = dP3 @ q(p["fc3.weight"])
    dmu = q(dz + KL * c["mu"] / n_k)
    dlv = q(dz * c["eps"] * 0.5 * c["std"] + KL * 0.5 * (np.exp(c["logvar"]) - 1.0) / n_k)
    dP1f = (dmu @ W21 + dlv @ W22) * (c["h1"] > 0)
    want = O.fp8_e4m3_round((dP1f * s_dp1).astype(np.float32))
    assert 8.0 < np.abs(want).max() <= 448.0, np.abs(want).max()          # the scale under test uses the format's range
    assert _rel_l2(dp1q, want) < 3e-2 and float((dp1q != want).mean()) < 0.2, (_rel_l2(dp1q, want), (dp1q != want).mean())
    # delayed scaling: unfreeze; the next step's first kernel turns the maximum measured above into dP1's scale
    e.set_fp8_scales(freeze_h3=False)
    e.step(torch.from_numpy(make_frames(B, S, 77)).cuda(), torch.from_numpy(make_eps(B, L, 78)).cuda())
    st2 = e.fp8_state()
    amax = float(np.abs(dP1f).max())
    assert abs(st2[14] / amax - 1.0) < 0.05, (st2[14], amax)
    assert abs(st2[13] * st2[14] / 224.0 - 1.0) < 1e-5 and abs(st2[15] * st2[13] * st2[0] - 1.0) < 1e-5
    assert np.isfinite(e.last_loss()[0])


def test_fp8_full_step_on_resident_waveform_equals_step_on_gathered_frames_at_c2():
    """The fp8 weight path at C2 (fc1's weight gradient on fp8 operands: neither bf16 copy of the frames nor of dP1 is
    written) fed from a resident waveform -- the gather-and-quantise cast kernel -- against gather -> `step`: the same
    bits, two steps (the second one runs on latched scales)."""
    from rawaudiovae_kelsey_amd import data as D
    S, H, L, B, hop = 1024, 2048, 64, 4096, 128
    wave = np.random.default_rng(5).uniform(-1, 1, (B + 40) * hop + S).astype(np.float32)
    d = D.DeviceAudio(wave, S, hop)
    idx = torch.randperm(len(d), generator=torch.Generator().manual_seed(1))[:B].contiguous().cuda()
    a, b = _engine(S, H, L, B, fp8="full", seed=2), _engine(S, H, L, B, fp8="full", seed=2)
    eps = torch.from_numpy(make_eps(B, L, 9)).cuda()
    for _ in range(2):
        a.step(d.gather(idx), eps)
        b.step_frames(d, idx, eps=eps)
    torch.cuda.synchronize()
    assert torch.equal(a.param, b.param) and a.losses(2) == b.losses(2)
    assert a.fp8_state()[13] == b.fp8_state()[13] != 56.0 * B * S       # dP1's scale has been latched from a measurement


def test_fp8_split_phase_calls_follow_the_forwards_decision():
    """A forward enqueued by a call of its own (RV_PHASE_FWD) is not a full local step: it writes the bf16 copy of the
    frames and latches no scale for dP1.  A later backward + Adam call -- which, taken alone, looks like a full local
    step -- must then run fc1's weight gradient on those bf16 operands, not on an fp8 image of dP1 whose scale was never
    prepared (round-4 advisor).  Checked at C2 against the single-call step from the same weights: the same loss bit
    for bit (the forward is the same), dP1's scale untouched and its fp8 image never written, every averaged gradient
    (exp_avg / 0.1) within the fp8 path's tolerance of the single-call step's, fc1.weight's equal to the oracle's with
    the bf16 backward's rounding points."""
    from rawaudiovae_kelsey_amd import engine as E
    S, H, L, B = 1024, 2048, 64, 4096
    x = torch.from_numpy(make_frames(B, S, 1234)).cuda()
    eps = torch.from_numpy(make_eps(B, L, 4321)).cuda()
    one = _engine(S, H, L, B, fp8="full")
    two = _engine(S, H, L, B, fp8="full")
    guess = 56.0 * B * S
    one.step(x, eps)
    two.step(x, eps, phases=E.PHASE_FWD)
    two.step(x, eps, phases=E.PHASE_BWD_A | E.PHASE_BWD_B | E.PHASE_ADAM)
    torch.cuda.synchronize()
    assert one.last_loss() == two.last_loss()
    assert two.fp8_state()[13] == guess                                   # never latched, never used
    assert not two.buffer("dP1q", torch.uint8, (-1,)).any()               # ... and its image never written
    assert two.buffer("dP1", torch.bfloat16, (-1,)).any()                 # the bf16 dP1 was
    for k in PARAM_NAMES:
        a = two.view(two.exp_avg, k).cpu().numpy().astype(np.float64)
        b = one.view(one.exp_avg, k).cpu().numpy().astype(np.float64)
        assert _rel_l2(a, b) < (6e-2 if k == "fc1.weight" else 1e-3), (k, _rel_l2(a, b))
