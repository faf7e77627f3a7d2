"""Seeded synthetic inputs (weights in the reference's layout and default init, frames, eps) and the algorithmic
FLOP count: shared by bench.py, the tools, smoke(), the golden-vector generator and the parity tests -- a neutral
module, so that bench.py does not reach into oracle/ for its workload (oracle/inputs.py re-exports it for the
tests).  numpy PCG64 only, so the GPU box regenerates the exact arrays the reference saw without needing torch's
CPU generator.

Shapes follow the reference: weights are nn.Linear `[out, in]` row-major fp32
(reference rawvae/model.py:13-17), default-initialised U(+-1/sqrt(fan_in))
for weight and bias alike.
"""
import numpy as np

PARAM_NAMES = (
    "fc1.weight", "fc1.bias",
    "fc21.weight", "fc21.bias",
    "fc22.weight", "fc22.bias",
    "fc3.weight", "fc3.bias",
    "fc4.weight", "fc4.bias",
)


def param_shapes(S, H, L):
    return {
        "fc1.weight": (H, S), "fc1.bias": (H,),
        "fc21.weight": (L, H), "fc21.bias": (L,),
        "fc22.weight": (L, H), "fc22.bias": (L,),
        "fc3.weight": (H, L), "fc3.bias": (H,),
        "fc4.weight": (S, H), "fc4.bias": (S,),
    }


def make_params(S, H, L, seed=0):
    """U(+-1/sqrt(fan_in)) weights and biases, float32, in PARAM_NAMES order."""
    rng = np.random.default_rng(seed)
    shapes = param_shapes(S, H, L)
    out = {}
    for name in PARAM_NAMES:
        shp = shapes[name]
        layer = name.split(".")[0]
        fan_in = shapes[layer + ".weight"][1]
        bound = 1.0 / np.sqrt(fan_in)
        out[name] = rng.uniform(-bound, bound, size=shp).astype(np.float32)
    return out


def make_frames(B, S, seed=1234):
    """Synthetic waveform frames, i.i.d. U(-1, 1) float32 (audio range)."""
    rng = np.random.default_rng(seed)
    return rng.uniform(-1.0, 1.0, size=(B, S)).astype(np.float32)


def make_eps(B, L, seed=4321):
    """Explicit N(0,1) draw that replaces `torch.randn_like(std)`
    (reference rawvae/model.py:25) in parity runs."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal(size=(B, L)).astype(np.float32)


def num_params(S, H, L):
    return sum(int(np.prod(s)) for s in param_shapes(S, H, L).values())


def flops_per_frame(S, H, L):
    """Algorithmic fwd+bwd FLOPs per frame: 10*S*H + 18*H*L."""
    return 10 * S * H + 18 * H * L
