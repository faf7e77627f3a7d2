"""numpy restatement of the reference training step (oracle; see oracle/__init__.py).

Every function cites the reference lines it follows.  Arithmetic runs in the
dtype asked for (`np.float64` for the tight pin, `np.float32` to mirror the
reference's fp32).  `quant="bf16"` additionally rounds the GEMM operands to
bfloat16 (round-to-nearest-even) at exactly the points where the HIP path
stores bf16, so kernel arithmetic can be checked far more tightly than the
bf16-vs-fp32 tolerance would allow.

Gradients are hand-derived and pinned against the reference's
autograd through the golden vectors.
"""
import numpy as np

from .inputs import PARAM_NAMES

ADAM_BETA1 = 0.9
ADAM_BETA2 = 0.999
ADAM_EPS = 1e-8


def bf16_round(a):
    """Round float array to bfloat16 (RNE), return in the input dtype."""
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    u = a32.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    out = u.astype(np.uint32).view(np.float32)
    return out.astype(a.dtype if hasattr(a, "dtype") else np.float32)


def fp8_e4m3_round(a):
    """Round to OCP fp8 e4m3 (4 exponent bits, bias 7; 3 mantissa bits; max 448, no infinities), round to
    nearest even, saturating; subnormals (step 2^-9 below 2^-6) kept.  Returns float64/32 values."""
    a = np.asarray(a)
    mag = np.minimum(np.abs(a.astype(np.float64)), 448.0)
    e = np.floor(np.log2(np.where(mag > 0, mag, 1.0)))
    e = np.maximum(e, -6.0)
    step = np.exp2(e - 3.0)
    q = np.rint(mag / step) * step          # np.rint rounds half to even
    q = np.minimum(q, 448.0)
    return (np.sign(a) * q).astype(a.dtype)


def _q(a, quant):
    return bf16_round(a) if quant in ("bf16", "fp8") else a


def _q8(a, scale):
    """fp8 operand as the GEMM sees it: fp8(a * scale) / scale."""
    return fp8_e4m3_round(np.asarray(a, dtype=np.float32) * np.float32(scale)).astype(np.float64) / scale


def cast_params(params, dtype):
    return {k: np.asarray(params[k], dtype=dtype) for k in PARAM_NAMES}


def encode(params, x, quant=None, fp8_scales=None):
    """h1 = relu(x W1^T + b1); mu, logvar = two heads on h1.
    Reference: rawvae/model.py:19-21.  quant="fp8": fc1's two operands are fp8(x * s_x), fp8(W1 * s_w1)."""
    if quant == "fp8":
        h1 = np.maximum((_q8(x, fp8_scales["x"]) @ _q8(params["fc1.weight"], fp8_scales["w1"]).T).astype(x.dtype)
                        + params["fc1.bias"], 0)
    else:
        xq = _q(x, quant)
        h1 = np.maximum(xq @ _q(params["fc1.weight"], quant).T + params["fc1.bias"], 0)
    h1 = _q(h1, quant)
    mu = h1 @ _q(params["fc21.weight"], quant).T + params["fc21.bias"]
    logvar = h1 @ _q(params["fc22.weight"], quant).T + params["fc22.bias"]
    return h1, mu, logvar


def reparameterize(mu, logvar, eps):
    """z = mu + eps * exp(0.5 logvar) with an explicit eps.
    Reference: rawvae/model.py:23-26 (eps there is torch.randn_like(std))."""
    std = np.exp(0.5 * logvar)
    return mu + eps * std, std


def decode(params, z, quant=None, fp8_scales=None):
    """h3 = relu(z W3^T + b3); recon = tanh(h3 W4^T + b4).
    Reference: rawvae/model.py:28-30.  quant="fp8": fc4 reads fp8(h3 * s_h3) (quantised from the fp32 value, as
    the producing epilogue does) and fp8(W4 * s_w4); h3 itself is kept in bf16 for the backward."""
    zq = _q(z, quant)
    h3f = np.maximum(zq @ _q(params["fc3.weight"], quant).T + params["fc3.bias"], 0)
    h3 = _q(h3f, quant)
    if quant == "fp8":
        h3q = _q8(h3f, fp8_scales["h3"])
        # the image keeps the ReLU mask: a positive activation is held at the smallest e4m3 subnormal instead of
        # quantising to zero (csrc/common.h fp8_keep_positive), so (h3q > 0) == (h3 > 0) for the fp8 backward's mask
        h3q = np.where((h3f > 0) & (h3q == 0), 2.0 ** -9 / fp8_scales["h3"], h3q)
        pre = (h3q @ _q8(params["fc4.weight"], fp8_scales["w4"]).T).astype(z.dtype)
        recon = np.tanh(pre + params["fc4.bias"])
        return h3, recon, h3q.astype(z.dtype)
    recon = np.tanh(h3 @ _q(params["fc4.weight"], quant).T + params["fc4.bias"])
    return h3, recon, None


def forward(params, x, eps, quant=None, fp8_scales=None):
    """Reference: rawvae/model.py:32-35.  Returns every intermediate.
    quant="fp8" (fp8_scales = {"x", "w1", "w4", "h3"}): the HIP path's fp8 mode -- fc1 and fc4 forward on
    e4m3 operands, everything else (and the whole backward) at the bf16 rounding points."""
    S = params["fc1.weight"].shape[1]
    x = x.reshape(-1, S)
    h1, mu, logvar = encode(params, x, quant, fp8_scales)
    z, std = reparameterize(mu, logvar, eps)
    h3, recon, h3q = decode(params, z, quant, fp8_scales)
    return dict(x=x, h1=h1, mu=mu, logvar=logvar, std=std, eps=eps, z=_q(z, quant), h3q=h3q,
                h3=h3, recon=recon)


def loss_function(recon, x, mu, logvar, kl_beta):
    """mse_loss(mean) + kl_beta * (-0.5 * mean(1 + logvar - mu^2 - exp(logvar))).
    Reference: rawvae/model.py:38-47."""
    x = x.reshape(recon.shape)
    mse = np.mean((recon - x) ** 2)
    kld = -0.5 * np.mean(1 + logvar - mu ** 2 - np.exp(logvar))
    return mse + kl_beta * kld, mse, kld


def backward(params, c, kl_beta, quant=None, fp8_scales=None):
    """Gradients of loss_function(forward(x)) w.r.t. the ten parameters.
    Stands in for `loss.backward()` (train.py:191); derivation SURVEY.md 3.4.
    quant="fp8" with fp8_scales (the forward's, plus "dp4"): the HIP path's fp8 backward of fc4 -- dP4 leaves the fc4
    forward's epilogue as fp8(dP4 * s_dp4) (from the fp32 value; the bias gradient is summed in fp32 before that), the
    dgrad multiplies it with fp8(W4 * s_w4), the wgrad with the fp8 image of h3 the fc3 forward wrote; everything else at
    the bf16 rounding points; with "dp1" as well, fc1's weight gradient on fp8 operands.  quant="fp8" without
    fp8_scales: bf16 rounding points throughout (the forward-only mode)."""
    x, h1, mu, logvar, std, eps, z, h3, recon = (
        c[k] for k in ("x", "h1", "mu", "logvar", "std", "eps", "z", "h3", "recon"))
    B, S = x.shape
    L = mu.shape[1]
    n_r = B * S
    n_k = B * L
    xq = _q(x, quant)
    W4, W3 = _q(params["fc4.weight"], quant), _q(params["fc3.weight"], quant)
    W21, W22 = _q(params["fc21.weight"], quant), _q(params["fc22.weight"], quant)
    g = {}
    if quant == "fp8" and fp8_scales is not None:
        dP4f = (2.0 / n_r) * (recon - x) * (1.0 - recon * recon)
        dP4 = _q8(dP4f, fp8_scales["dp4"])
        g["fc4.weight"] = dP4.T @ c["h3q"]
        g["fc4.bias"] = dP4f.sum(0)
        dP3 = _q((dP4 @ _q8(params["fc4.weight"], fp8_scales["w4"])) * (h3 > 0), quant)
    else:
        dP4 = _q((2.0 / n_r) * (recon - x) * (1.0 - recon * recon), quant)
        g["fc4.weight"] = dP4.T @ h3
        g["fc4.bias"] = dP4.sum(0)
        dP3 = _q((dP4 @ W4) * (h3 > 0), quant)
    g["fc3.weight"] = dP3.T @ z
    g["fc3.bias"] = dP3.sum(0)
    dz = dP3 @ W3
    dmu = _q(dz + kl_beta * mu / n_k, quant)
    dlv = _q(dz * eps * 0.5 * std + kl_beta * 0.5 * (np.exp(logvar) - 1.0) / n_k, quant)
    g["fc21.weight"] = dmu.T @ h1
    g["fc21.bias"] = dmu.sum(0)
    g["fc22.weight"] = dlv.T @ h1
    g["fc22.bias"] = dlv.sum(0)
    if quant == "fp8" and fp8_scales is not None and "dp1" in fp8_scales:
        # fc1's fp8 weight gradient (the full local step of the fp8 weight path): dP1 leaves the heads' backward as
        # fp8(dP1 * s_dp1) from the fp32 value (fc1's bias gradient is summed in fp32 before that), the GEMM multiplies
        # it with the fp8 image of x fc1's forward read
        dP1f = (dmu @ W21 + dlv @ W22) * (h1 > 0)
        g["fc1.weight"] = _q8(dP1f, fp8_scales["dp1"]).T @ _q8(x, fp8_scales["x"])
        g["fc1.bias"] = dP1f.sum(0)
        return g
    dP1 = _q((dmu @ W21 + dlv @ W22) * (h1 > 0), quant)
    g["fc1.weight"] = dP1.T @ xq
    g["fc1.bias"] = dP1.sum(0)
    return g


def adam_init(params):
    return {"step": 0,
            "exp_avg": {k: np.zeros_like(params[k]) for k in PARAM_NAMES},
            "exp_avg_sq": {k: np.zeros_like(params[k]) for k in PARAM_NAMES}}


def adam_step(params, grads, state, lr):
    """torch.optim.Adam defaults (betas 0.9/0.999, eps 1e-8, no weight decay,
    no amsgrad) as constructed at train.py:163 and stepped at train.py:193.
    In-place on params/state; same operation order as torch's single-tensor path:
    denom = sqrt(v)/sqrt(1-b2^t) + eps ; p -= (lr/(1-b1^t)) * m/denom."""
    state["step"] += 1
    t = state["step"]
    bc1 = 1.0 - ADAM_BETA1 ** t
    bc2 = 1.0 - ADAM_BETA2 ** t
    for k in PARAM_NAMES:
        g = grads[k]
        m = state["exp_avg"][k]
        v = state["exp_avg_sq"][k]
        m *= ADAM_BETA1
        m += (1.0 - ADAM_BETA1) * g
        v *= ADAM_BETA2
        v += (1.0 - ADAM_BETA2) * g * g
        denom = np.sqrt(v) / np.sqrt(bc2) + ADAM_EPS
        params[k] -= (lr / bc1) * (m / denom)
    return params, state


def train_step(params, state, x, eps, kl_beta, lr, quant=None):
    """zero_grad -> forward -> loss -> backward -> Adam, train.py:184-193."""
    c = forward(params, x, eps, quant)
    loss, mse, kld = loss_function(c["recon"], c["x"], c["mu"], c["logvar"], kl_beta)
    grads = backward(params, c, kl_beta, quant)
    adam_step(params, grads, state, lr)
    return loss, c, grads


def frame_count(n_samples, segment_length, hop):
    """AudioDataset.__len__ after padding to a multiple of hop
    (rawvae/dataset.py:99-104,121)."""
    if segment_length % hop != 0:
        raise ValueError("segment_length {} is not a multiple of hop_size {}".format(
            segment_length, hop))
    padded = n_samples if n_samples % hop == 0 else n_samples + hop - n_samples % hop
    return padded // hop - segment_length // hop + 1, padded


def hop_frames(audio, segment_length, hop, indices=None):
    """AudioDataset.__getitem__ for a set of indices (rawvae/dataset.py:108-118)."""
    n, padded = frame_count(len(audio), segment_length, hop)
    a = np.zeros(padded, dtype=audio.dtype)
    a[:len(audio)] = audio
    if indices is None:
        indices = np.arange(n)
    idx = np.asarray(indices)[:, None] * hop + np.arange(segment_length)[None, :]
    return a[idx]


def eval_frames(audio, segment_length):
    """TestDataset: non-overlapping frames, tail zero-padded
    (rawvae/dataset.py:141-160)."""
    n = len(audio)
    padded = n if n % segment_length == 0 else n + segment_length - n % segment_length
    a = np.zeros(padded, dtype=audio.dtype)
    a[:n] = audio
    return a.reshape(-1, segment_length)
