"""numpy restatement of the DEEP variant (BASELINE.json configs[3]: `depth` hidden Linear+ReLU
layers on each side instead of one).  The reference has no such model: this is the
build's own definition -- the same arithmetic as oracle/vae_oracle.py applied layer by layer --
and with depth=1 it reduces exactly to the reference topology (checked in tests).

Parameter naming: enc.{i}.weight/bias (i < depth), fc21, fc22, dec.{i}.weight/bias, fc4.
"""
import numpy as np

from .vae_oracle import _q, loss_function  # noqa: F401  (same loss as the reference)


def param_names(depth):
    n = []
    for i in range(depth):
        n += ["enc.%d.weight" % i, "enc.%d.bias" % i]
    n += ["fc21.weight", "fc21.bias", "fc22.weight", "fc22.bias"]
    for i in range(depth):
        n += ["dec.%d.weight" % i, "dec.%d.bias" % i]
    n += ["fc4.weight", "fc4.bias"]
    return n


def param_shapes(S, H, L, depth):
    sh = {}
    for i in range(depth):
        sh["enc.%d.weight" % i] = (H, S if i == 0 else H)
        sh["enc.%d.bias" % i] = (H,)
        sh["dec.%d.weight" % i] = (H, L if i == 0 else H)
        sh["dec.%d.bias" % i] = (H,)
    sh.update({"fc21.weight": (L, H), "fc21.bias": (L,), "fc22.weight": (L, H), "fc22.bias": (L,),
               "fc4.weight": (S, H), "fc4.bias": (S,)})
    return sh


def make_params(S, H, L, depth, seed=0):
    """nn.Linear default init U(+-1/sqrt(fan_in)), numpy PCG64, in param_names() order."""
    rng = np.random.default_rng(seed)
    sh = param_shapes(S, H, L, depth)
    out = {}
    for name in param_names(depth):
        layer = name.rsplit(".", 1)[0]
        bound = 1.0 / np.sqrt(sh[layer + ".weight"][1])
        out[name] = rng.uniform(-bound, bound, size=sh[name]).astype(np.float32)
    return out


def forward(p, x, eps, depth, quant=None):
    a = _q(x, quant)
    enc = []
    for i in range(depth):
        a = _q(np.maximum(a @ _q(p["enc.%d.weight" % i], quant).T + p["enc.%d.bias" % i], 0), quant)
        enc.append(a)
    mu = a @ _q(p["fc21.weight"], quant).T + p["fc21.bias"]
    logvar = a @ _q(p["fc22.weight"], quant).T + p["fc22.bias"]
    std = np.exp(0.5 * logvar)
    z = _q(mu + eps * std, quant)
    a = z
    dec = []
    for i in range(depth):
        a = _q(np.maximum(a @ _q(p["dec.%d.weight" % i], quant).T + p["dec.%d.bias" % i], 0), quant)
        dec.append(a)
    recon = np.tanh(a @ _q(p["fc4.weight"], quant).T + p["fc4.bias"])
    return dict(x=x, enc=enc, mu=mu, logvar=logvar, std=std, eps=eps, z=z, dec=dec, recon=recon)


def backward(p, c, kl_beta, depth, quant=None):
    x, mu, logvar, std, eps, z, recon = (c[k] for k in ("x", "mu", "logvar", "std", "eps", "z", "recon"))
    B, S = x.shape
    L = mu.shape[1]
    n_r, n_k = B * S, B * L
    g = {}
    dy = _q((2.0 / n_r) * (recon - x) * (1.0 - recon * recon), quant)
    g["fc4.weight"] = dy.T @ c["dec"][-1]
    g["fc4.bias"] = dy.sum(0)
    w = _q(p["fc4.weight"], quant)
    for i in range(depth - 1, -1, -1):
        a_out = c["dec"][i]
        dy = _q((dy @ w) * (a_out > 0), quant)
        a_in = c["dec"][i - 1] if i > 0 else z
        g["dec.%d.weight" % i] = dy.T @ a_in
        g["dec.%d.bias" % i] = dy.sum(0)
        w = _q(p["dec.%d.weight" % i], quant)
    dz = dy @ w
    dmu = _q(dz + kl_beta * mu / n_k, quant)
    dlv = _q(dz * eps * 0.5 * std + kl_beta * 0.5 * (np.exp(logvar) - 1.0) / n_k, quant)
    h = c["enc"][-1]
    g["fc21.weight"], g["fc21.bias"] = dmu.T @ h, dmu.sum(0)
    g["fc22.weight"], g["fc22.bias"] = dlv.T @ h, dlv.sum(0)
    dy = _q((dmu @ _q(p["fc21.weight"], quant) + dlv @ _q(p["fc22.weight"], quant)) * (h > 0), quant)
    for i in range(depth - 1, -1, -1):
        a_in = c["enc"][i - 1] if i > 0 else _q(x, quant)
        g["enc.%d.weight" % i] = dy.T @ a_in
        g["enc.%d.bias" % i] = dy.sum(0)
        if i > 0:
            dy = _q((dy @ _q(p["enc.%d.weight" % i], quant)) * (a_in > 0), quant)
    return g
