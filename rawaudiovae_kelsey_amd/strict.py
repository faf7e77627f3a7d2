"""Strict-fp32 training step: the reference's inner loop (train.py:184-193 over rawvae/model.py:19-47) with EVERY
product in exact fp32 (`rv_linear_fp32`: v_mfma_f32_32x32x2_f32, a k-ordered fp32 fma chain) and fp32
elementwise kernels -- no bf16 anywhere.  Not a fast path (f32 MFMA runs at 1/16 of the bf16 rate and the
transposed operands of the backward products are materialised); it exists so that gradients, Adam state and
trajectories of the HIP formulas can be held to the reference at fp32 tolerances (SURVEY 8d strict gate: 1e-5),
which the bf16 path's ReLU-mask flips do not allow.

PyTorch supplies memory, transposed copies and the step counter; all model arithmetic runs in the C-ABI kernels.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_TANH, ParamDesc, lib, ptr, stream_ptr
from .engine import PARAM_NAMES, param_shapes
from .ops import _colsum


class StrictFp32Engine:
    def __init__(self, segment_length, n_units, latent_dim, device="cuda", kl_beta=1e-4, lr=1e-4):
        self.S, self.H, self.L = int(segment_length), int(n_units), int(latent_dim)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RvError("StrictFp32Engine needs a GPU device; there is no CPU path")
        self.kl_beta, self.lr = float(kl_beta), float(lr)
        self.shapes = param_shapes(self.S, self.H, self.L)
        self.offsets, o = {}, 0
        for k in PARAM_NAMES:
            self.offsets[k] = o
            n = 1
            for d in self.shapes[k]:
                n *= d
            o += n
        self.n_params = o
        f32 = dict(dtype=torch.float32, device=self.device)
        self.param, self.exp_avg, self.exp_avg_sq, self.grad = (torch.zeros(o, **f32) for _ in range(4))
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.loss = torch.zeros(4, **f32)
        self._loss_ws = torch.zeros(lib().rv_loss_fused_workspace_bytes(), dtype=torch.uint8, device=self.device)

    def view(self, arena, name):
        n = 1
        for d in self.shapes[name]:
            n *= d
        return arena[self.offsets[name]:self.offsets[name] + n].view(self.shapes[name])

    def load_params(self, params):
        with torch.no_grad():
            for k in PARAM_NAMES:
                self.view(self.param, k).copy_(torch.as_tensor(params[k]).to(self.device, torch.float32))

    # ---- kernels ----
    def _lin(self, x, W, b=None, act=ACT_NONE):
        """act(x W^T + b): x [M, K], W [N, K] contiguous fp32."""
        M, K = x.shape
        N = W.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=self.device)
        lib().rv_linear_fp32(ptr(x), K, ptr(W), K, ptr(b), M, N, K, int(act), ptr(y), N, stream_ptr())
        return y

    def _ew(self, op, a, b):
        out = torch.empty_like(a)
        lib().rv_ew_f32(op, ptr(a), ptr(b), a.numel(), ptr(out), stream_ptr())
        return out

    def forward(self, x, eps):
        p = {k: self.view(self.param, k) for k in PARAM_NAMES}
        c = {"x": x}
        c["h1"] = self._lin(x, p["fc1.weight"], p["fc1.bias"], ACT_RELU)
        c["mu"] = self._lin(c["h1"], p["fc21.weight"], p["fc21.bias"])
        c["logvar"] = self._lin(c["h1"], p["fc22.weight"], p["fc22.bias"])
        c["z"] = torch.empty_like(c["mu"])
        lib().rv_reparameterize(ptr(c["mu"]), ptr(c["logvar"]), c["mu"].numel(), ptr(eps), None, 0, 0, ptr(c["z"]),
                                stream_ptr())
        c["h3"] = self._lin(c["z"], p["fc3.weight"], p["fc3.bias"], ACT_RELU)
        c["recon"] = self._lin(c["h3"], p["fc4.weight"], p["fc4.bias"], ACT_TANH)
        c["eps"] = eps
        return c

    def backward(self, c):
        """Loss (model.py:38-47) and the ten gradients (train.py:191) into the flat `grad` arena."""
        x, B = c["x"], c["x"].shape[0]
        p = {k: self.view(self.param, k) for k in PARAM_NAMES}
        g = {k: self.view(self.grad, k) for k in PARAM_NAMES}
        d_recon, d_mu, d_lv = torch.empty_like(c["recon"]), torch.empty_like(c["mu"]), torch.empty_like(c["mu"])
        lib().rv_loss_fused(ptr(c["recon"]), ptr(x), ptr(c["mu"]), ptr(c["logvar"]), B, self.S, self.L, self.kl_beta,
                            ptr(self.loss), ptr(d_recon), ptr(d_mu), ptr(d_lv), ptr(self._loss_ws), stream_ptr())
        T = lambda t: t.t().contiguous()   # noqa: E731  (data movement only)
        dP4 = self._ew(0, d_recon, c["recon"])
        g["fc4.weight"].copy_(self._lin(T(dP4), T(c["h3"])))
        g["fc4.bias"].copy_(_colsum(dP4, False, B, self.S, self.S))
        dP3 = self._ew(1, self._lin(dP4, T(p["fc4.weight"])), c["h3"])
        g["fc3.weight"].copy_(self._lin(T(dP3), T(c["z"])))
        g["fc3.bias"].copy_(_colsum(dP3, False, B, self.H, self.H))
        dz = self._lin(dP3, T(p["fc3.weight"]))
        dmu_r, dlv_r = torch.empty_like(dz), torch.empty_like(dz)
        lib().rv_reparameterize_bwd(ptr(dz), ptr(c["eps"]), ptr(c["logvar"]), dz.numel(), ptr(dmu_r), ptr(dlv_r),
                                    stream_ptr())
        dmu, dlv = self._ew(2, dmu_r, d_mu), self._ew(2, dlv_r, d_lv)
        h1T = T(c["h1"])
        g["fc21.weight"].copy_(self._lin(T(dmu), h1T))
        g["fc21.bias"].copy_(_colsum(dmu, False, B, self.L, self.L))
        g["fc22.weight"].copy_(self._lin(T(dlv), h1T))
        g["fc22.bias"].copy_(_colsum(dlv, False, B, self.L, self.L))
        dh1 = self._ew(2, self._lin(dmu, T(p["fc21.weight"])), self._lin(dlv, T(p["fc22.weight"])))
        dP1 = self._ew(1, dh1, c["h1"])
        c["dP1"] = dP1   # kept for the per-element summation-error check of tests/test_strict_fp32_gpu.py
        g["fc1.weight"].copy_(self._lin(T(dP1), T(x)))
        g["fc1.bias"].copy_(_colsum(dP1, False, B, self.H, self.H))

    def adam(self):
        self.step_counter += 1
        d = (ParamDesc * 10)()
        for i, k in enumerate(PARAM_NAMES):
            shp = self.shapes[k]
            rows, cols = (shp[0], shp[1]) if len(shp) == 2 else (1, shp[0])
            d[i] = ParamDesc(self.offsets[k], rows, cols, self.grad.data_ptr() + 4 * self.offsets[k], cols, 0, 1, None, None, 0)
        lib().rv_adam_multi(d, 10, ptr(self.param), ptr(self.exp_avg), ptr(self.exp_avg_sq), None, None, self.lr, 1.0,
                            ptr(self.step_counter), stream_ptr())

    def step(self, x, eps):
        """zero_grad / forward / loss / backward / Adam.step; returns the forward's intermediates."""
        c = self.forward(x, eps)
        self.backward(c)
        self.adam()
        return c

    def last_loss(self):
        return tuple(float(v) for v in self.loss[:3].tolist())
