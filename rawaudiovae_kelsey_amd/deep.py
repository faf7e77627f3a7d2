"""Deep variant of the raw-audio VAE (BASELINE.json configs[3]: 2048-sample frames, 4-layer
encoder/decoder, latent 256): `depth` Linear+ReLU layers on each side instead of one.

The reference has no such model -- this is a build extension that shows the
kernel family is not specialised to the five-Linear topology.  The training step is sequenced
from Python over the same C-ABI entry points the fused plan uses (`rv_linear_fwd`,
`rv_linear_dgrad_wgrad` -- paired 256x256 launch for every H x H layer --, `rv_latent_fwd` / `rv_latent_bwd`
(the latent-sized GEMMs with the reparameterisation in their epilogues, round 6),
`rv_decode_out_loss_fwd`, `rv_adam_multi`); capture it in a hipGraph (`engine.Graph`) to remove
the per-launch host cost.  With depth=1 it computes exactly what `TrainEngine` computes.

Parity is against the build's own CPU restatement (`oracle/deep_oracle.py`).
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import (ACT_RELU, SLAB_F16, SLAB_F32, TILE_256x256, ParamDesc, dgrad_wgrad_pick, gemm_pick, gemm_tile, lib, pad_dims, ptr,
                   stream_ptr)


def param_names(depth):
    n = []
    for i in range(depth):
        n += ["enc.%d.weight" % i, "enc.%d.bias" % i]
    n += ["fc21.weight", "fc21.bias", "fc22.weight", "fc22.bias"]
    for i in range(depth):
        n += ["dec.%d.weight" % i, "dec.%d.bias" % i]
    n += ["fc4.weight", "fc4.bias"]
    return n


class DeepVAE(nn.Module):
    """`VAE` with `depth` hidden layers per side; depth=1 is the reference topology."""

    def __init__(self, segment_length, n_units, latent_dim, depth=3):
        super().__init__()
        self.segment_length, self.n_units, self.latent_dim, self.depth = segment_length, n_units, latent_dim, depth
        self.enc = nn.ModuleList([nn.Linear(segment_length if i == 0 else n_units, n_units) for i in range(depth)])
        self.fc21 = nn.Linear(n_units, latent_dim)
        self.fc22 = nn.Linear(n_units, latent_dim)
        self.dec = nn.ModuleList([nn.Linear(latent_dim if i == 0 else n_units, n_units) for i in range(depth)])
        self.fc4 = nn.Linear(n_units, segment_length)

    def engine(self, batch_size, kl_beta, lr, seed=0, ring=256, slab_dtype="fp16"):
        eng = DeepTrainEngine(self.segment_length, self.n_units, self.latent_dim, self.depth, batch_size,
                              device=self.fc4.weight.device, kl_beta=kl_beta, lr=lr, seed=seed, ring=ring,
                              slab_dtype=slab_dtype)
        eng.adopt(self)
        return eng


class DeepTrainEngine:
    """Whole-step engine for `DeepVAE`, sequenced in Python over the C ABI."""

    def __init__(self, S, H, L, depth, batch_size, device="cuda", kl_beta=1e-4, lr=1e-4, seed=0, ring=256,
                 slab_dtype="fp16"):
        """slab_dtype: element type of the split-K slabs of the LARGE weight gradients (every layer whose padded extents
        are multiples of 32 and 8: the H x H layers, the first and the last) -- "fp16" (default, as TrainEngine's):
        block-floating-point fp16 with one power-of-two scale per 32 x 32 granule and slab (half the bytes the
        weight-gradient GEMMs write and the optimizer reads back: 200 of the step's ~900 MB of optimizer-side traffic at
        the C4 shape); "fp32".  The latent-sized gradients (heads, dec.0) keep fp32 slabs."""
        self.S, self.H, self.L, self.depth, self.B = int(S), int(H), int(L), int(depth), int(batch_size)
        if slab_dtype not in ("fp16", "fp32"):
            raise _lib.RvError("slab_dtype %r (expected 'fp16' or 'fp32')" % (slab_dtype,))
        self.slab_dtype = slab_dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RvError("DeepTrainEngine needs a GPU device; there is no CPU path")
        self.kl_beta, self.lr, self.seed, self.ring = float(kl_beta), float(lr), int(seed), int(ring)
        self.Bp, self.Sp, self.Hp, self.Lp = pad_dims(self.B, self.S, self.H, self.L)
        Bp, Sp, Hp, Lp, d = self.Bp, self.Sp, self.Hp, self.Lp, self.depth
        dev = self.device
        self.names = param_names(d)
        self.shapes = {}
        for i in range(d):
            self.shapes["enc.%d.weight" % i] = (H, S if i == 0 else H)
            self.shapes["enc.%d.bias" % i] = (H,)
            self.shapes["dec.%d.weight" % i] = (H, L if i == 0 else H)
            self.shapes["dec.%d.bias" % i] = (H,)
        self.shapes.update({"fc21.weight": (L, H), "fc21.bias": (L,), "fc22.weight": (L, H), "fc22.bias": (L,),
                            "fc4.weight": (S, H), "fc4.bias": (S,)})
        self.offsets, o = {}, 0
        for k in self.names:
            self.offsets[k] = o
            n = 1
            for v in self.shapes[k]:
                n *= v
            o += n
        self.n_params = o
        f32 = dict(dtype=torch.float32, device=dev)
        bf = dict(dtype=torch.bfloat16, device=dev)
        self.param, self.exp_avg, self.exp_avg_sq = (torch.zeros(o, **f32) for _ in range(3))
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.loss_ring = torch.zeros(self.ring, 4, **f32)

        def z_(*shape, **kw):
            return torch.zeros(*shape, **kw)
        # forward operands
        self.xb = z_(Bp, Sp, **bf)
        self.enc_act = [z_(Bp, Hp, **bf) for _ in range(d)]
        self.dec_act = [z_(Bp, Hp, **bf) for _ in range(d)]
        self.mulv = z_(Bp, 2 * Lp, **f32)
        self.eps = z_(Bp * Lp, **f32)
        self.z = z_(Bp, Lp, **bf)
        self.n_kl = Bp * Lp // 1024
        self.kl_part = z_(self.n_kl, **f32)
        bm_o, bn_o = gemm_tile(Bp, Sp, 1)
        self.n_mse = (Bp // bm_o) * (Sp // bn_o)
        self.mse_part = z_(self.n_mse, **f32)
        # backward operands
        self.dP_out = z_(Bp, Sp, **bf)
        self.d_dec = [z_(Bp, Hp, **bf) for _ in range(d)]
        self.d_enc = [z_(Bp, Hp, **bf) for _ in range(d)]
        self.dmulv = z_(Bp, 2 * Lp, **bf)
        # per-tensor shadows, gradient slabs and bias-gradient partials
        self.shadow, self.slabs, self.splits, self.bias_part = {}, {}, {}, {}
        self.unscale = {}      # name -> [splits, rows_p / 32, cols_p / 32] fp32: 2^-e per granule of an fp16 slab

        def weight(name, rows_p, cols_p):
            self.shadow[name] = z_(rows_p, cols_p, **bf)

        def bias(name, n_p):
            self.shadow[name] = z_(n_p, **f32)
        for i in range(d):
            weight("enc.%d.weight" % i, Hp, Sp if i == 0 else Hp)
            bias("enc.%d.bias" % i, Hp)
            weight("dec.%d.weight" % i, Hp, Lp if i == 0 else Hp)
            bias("dec.%d.bias" % i, Hp)
        self.Whb = z_(2 * Lp, Hp, **bf)
        self.bhp = z_(2 * Lp, **f32)
        weight("fc4.weight", Sp, Hp)
        bias("fc4.bias", Sp)
        # layer backward plans: (paired, bm of the dgrad that produces the input-side dY, wgrad splits)
        self.plan_out = dgrad_wgrad_pick(Bp, Hp, Sp)            # fc4:   dy [Bp,Sp], W [Sp,Hp]
        self.plan_hh = dgrad_wgrad_pick(Bp, Hp, Hp)             # H x H layers
        self.plan_heads = dgrad_wgrad_pick(Bp, Hp, 2 * Lp)       # heads: dy [Bp,2Lp], W [2Lp,Hp]

        def slab(name, splits, rows_p, cols_p, half_ok=False):
            self.splits[name] = splits
            if half_ok and self.slab_dtype == "fp16" and rows_p % 32 == 0 and cols_p % 32 == 0:
                self.slabs[name] = z_(splits, rows_p, cols_p, dtype=torch.float16, device=dev)
                self.unscale[name] = z_(splits, rows_p // 32, cols_p // 32, **f32)
            else:
                self.slabs[name] = z_(splits, rows_p, cols_p, **f32)
        slab("fc4.weight", self.plan_out[2], Sp, Hp, True)
        for i in range(1, d):
            slab("dec.%d.weight" % i, self.plan_hh[2], Hp, Hp, True)
            slab("enc.%d.weight" % i, self.plan_hh[2], Hp, Hp, True)
        slab("dec.0.weight", gemm_pick(Hp, Lp, Bp)[2], Hp, Lp)
        # The first layer's weight gradient dW = dY^T x is the last GEMM of the backward and has no dgrad to share a launch
        # with: on 256 x 256 tiles with the ping-pong loop (the wgrad half of the paired launches, alone) and as many K
        # splits as fill the chip -- 64 tiles x 4 at the C4 shape, 31 us -- where the extents tile; the picker's
        # 256 x 128 tiles with two transposed operands took 55 us there (profiles/r05_deep_kernel_stats.txt).
        sp0, self.tile_enc0 = gemm_pick(Hp, Sp, Bp)[2], -1
        if Hp % 256 == 0 and Sp % 256 == 0:
            tiles, kt = (Hp // 256) * (Sp // 256), Bp // 64
            while tiles * sp0 < 256 and kt % (2 * sp0) == 0 and kt // (2 * sp0) >= 2 and sp0 < 8:
                sp0 *= 2
            if kt % sp0 == 0:
                self.tile_enc0 = TILE_256x256
        slab("enc.0.weight", sp0, Hp, Sp, True)
        # (the heads' eight [2 Lp, Hp] slabs are block-floating-point fp16 too where their backward is the paired 256 x 256
        # launch -- round 6, as TrainEngine at the reference's latent width: 32 MB as fp32, written there and read by Adam)
        slab("heads.weight", self.plan_heads[2], 2 * Lp, Hp, bool(self.plan_heads[0]))
        # bias-gradient partial rows: produced by the kernel that creates the layer's dY
        self.bias_part["fc4.bias"] = z_(Bp // bm_o, Sp, **f32)
        self.bias_part["dec.%d.bias" % (d - 1)] = z_(Bp // self.plan_out[1], Hp, **f32)
        self.bias_part["enc.%d.bias" % (d - 1)] = z_(Bp // self.plan_heads[1], Hp, **f32)
        for i in range(d - 1):
            self.bias_part["dec.%d.bias" % i] = z_(Bp // self.plan_hh[1], Hp, **f32)
            self.bias_part["enc.%d.bias" % i] = z_(Bp // self.plan_hh[1], Hp, **f32)
        self.dbh_part = z_(Bp // 16, 2 * Lp, **f32)
        self._descs = self._build_descs()
        n = len(self.names)   # rv_adam_multi takes at most 16 tensors per launch
        self._chunks = [(ParamDesc * min(16, n - lo))(*[self._descs[j] for j in range(lo, min(lo + 16, n))])
                        for lo in range(0, n, 16)]
        self.host_steps = 0

    # ---- parameters -----------------------------------------------------
    def view(self, arena, name):
        n = 1
        for v in self.shapes[name]:
            n *= v
        o = self.offsets[name]
        return arena[o:o + n].view(self.shapes[name])

    def param_views(self):
        return {k: self.view(self.param, k) for k in self.names}

    def load_params(self, params):
        with torch.no_grad():
            for k in self.names:
                src = params[k]
                if not torch.is_tensor(src):
                    src = torch.as_tensor(src)
                self.view(self.param, k).copy_(src.to(self.device, torch.float32))
        self.refresh_shadows()

    def adopt(self, module):
        sd = dict(module.named_parameters())
        with torch.no_grad():
            for k in self.names:
                v = self.view(self.param, k)
                v.copy_(sd[k].detach().to(self.device, torch.float32))
                sd[k].data = v
        self.refresh_shadows()

    def _shadow_of(self, name):
        """(bf16 shadow ptr, f32 shadow ptr, ld) of a tensor (heads live inside the fused [2Lp,Hp] weight)."""
        Lp, Hp = self.Lp, self.Hp
        if name == "fc21.weight":
            return self.Whb.data_ptr(), None, Hp
        if name == "fc22.weight":
            return self.Whb.data_ptr() + 2 * Lp * Hp, None, Hp
        if name == "fc21.bias":
            return None, self.bhp.data_ptr(), 2 * Lp
        if name == "fc22.bias":
            return None, self.bhp.data_ptr() + 4 * Lp, 2 * Lp
        t = self.shadow[name]
        if name.endswith("weight"):
            return t.data_ptr(), None, t.shape[1]
        return None, t.data_ptr(), t.numel()

    def refresh_shadows(self):
        L_ = lib()
        st = stream_ptr()
        self.bhp.zero_()
        for k in self.names:
            src = self.view(self.param, k)
            sb, sf, ld = self._shadow_of(k)
            if k.endswith("weight"):
                rows_p = self.Lp if k in ("fc21.weight", "fc22.weight") else self.shadow[k].shape[0]
                L_.rv_cast_pad_bf16(ptr(src), src.shape[0], src.shape[1], src.shape[1], sb, rows_p, ld, ld, None, st)
            elif k in ("fc21.bias", "fc22.bias"):
                off = 0 if k == "fc21.bias" else self.Lp
                self.bhp[off:off + src.numel()].copy_(src)
            else:
                self.shadow[k].zero_()
                self.shadow[k][:src.numel()].copy_(src)

    def _build_descs(self):
        Lp, Hp = self.Lp, self.Hp
        descs = (ParamDesc * len(self.names))()
        for i, k in enumerate(self.names):
            rows, cols = (self.shapes[k] if len(self.shapes[k]) == 2 else (1, self.shapes[k][0]))
            sb, sf, ld = self._shadow_of(k)
            if k in ("fc21.weight", "fc22.weight"):
                sl = self.slabs["heads.weight"]
                base = sl.data_ptr() + (sl.element_size() * Lp * Hp if k == "fc22.weight" else 0)
                g = (base, Hp, 2 * Lp * Hp, self.splits["heads.weight"])
            elif k in ("fc21.bias", "fc22.bias"):
                base = self.dbh_part.data_ptr() + (4 * Lp if k == "fc22.bias" else 0)
                g = (base, 2 * Lp, 2 * Lp, self.dbh_part.shape[0])
            elif k.endswith("weight"):
                sl = self.slabs[k]
                g = (sl.data_ptr(), sl.shape[2], sl.shape[1] * sl.shape[2], sl.shape[0])
            else:
                bp = self.bias_part[k]
                g = (bp.data_ptr(), bp.shape[1], bp.shape[1], bp.shape[0])
            descs[i] = ParamDesc(self.offsets[k], rows, cols, g[0], g[1], g[2], g[3], sb, sf, ld)
            us = self.unscale.get("heads.weight" if k in ("fc21.weight", "fc22.weight") else k)
            if us is not None:     # fp16 slabs: the optimizer multiplies each slab value by its granule's 2^-e
                descs[i].grad_half = 1
                descs[i].grad_unscale = us.data_ptr() + (4 * (Lp // 32) * us.shape[2] if k == "fc22.weight" else 0)
                descs[i].us_ld, descs[i].us_split_stride = us.shape[2], us.shape[1] * us.shape[2]
        return descs

    def _slab_args(self, name):
        """(slab_dtype, unscale table) of a weight's gradient slabs, as rv_linear_dgrad_wgrad / rv_linear_wgrad take them."""
        us = self.unscale.get(name)
        return (SLAB_F16, us.data_ptr()) if us is not None else (SLAB_F32, None)

    # ---- one training step ---------------------------------------------
    def step(self, x, eps=None, recon_out=None, adam=True, stream=None):
        if x.dtype != torch.float32 or not x.is_contiguous() or x.numel() != self.B * self.S:
            raise _lib.RvError("step: x must be contiguous fp32 [B, S]")
        L_, st = lib(), stream_ptr(stream)
        # (plain stores throughout: the next layer's launch finds its input rows in the L2 that wrote them -- write-through
        # everywhere cost this engine 858 -> 951 us in round 3, and write-through for the split-K slabs alone, which
        # nothing reads before the optimizer, 848 -> 858-864 us in round 5: profiles/r05_deep.txt)
        self._enqueue(L_, st, x, eps, recon_out, adam)
        self.host_steps += 1

    def _enqueue(self, L_, st, x, eps, recon_out, adam):
        B, S, L, Bp, Sp, Hp, Lp, d = self.B, self.S, self.L, self.Bp, self.Sp, self.Hp, self.Lp, self.depth
        W = lambda k: ptr(self.shadow[k])  # noqa: E731
        ctr = ptr(self.step_counter)
        # forward
        L_.rv_cast_pad_bf16(ptr(x), B, S, S, ptr(self.xb), Bp, Sp, Sp, ctr, st)
        a, ka = self.xb, Sp
        for i in range(d):
            L_.rv_linear_fwd(ptr(a), ka, W("enc.%d.weight" % i), ka, W("enc.%d.bias" % i), Bp, Hp, ka, ACT_RELU,
                             ptr(self.enc_act[i]), Hp, st)
            a, ka = self.enc_act[i], Hp
        # heads + reparameterisation + KL partials + the first decoder layer: rv_latent_fwd (round 6: at this variant's
        # latent width of 256 the heads GEMM with the reparameterisation in its epilogue, then dec.0's forward GEMM --
        # no fp32 slabs of mu | logvar, no reparameterisation launch; csrc/latent.hip)
        L_.rv_latent_fwd(ptr(a), Hp, ptr(self.Whb), Hp, ptr(self.bhp), W("dec.0.weight"), Lp, W("dec.0.bias"), Bp, Hp, Lp,
                         B, L, ptr(eps), ptr(self.eps), self.seed, ctr, ptr(self.mulv), ptr(self.z), ptr(self.kl_part),
                         ptr(self.dec_act[0]), Hp, st)
        a, ka = self.dec_act[0], Hp
        for i in range(1, d):
            L_.rv_linear_fwd(ptr(a), ka, W("dec.%d.weight" % i), ka, W("dec.%d.bias" % i), Bp, Hp, ka, ACT_RELU,
                             ptr(self.dec_act[i]), Hp, st)
            a, ka = self.dec_act[i], Hp
        L_.rv_decode_out_loss_fwd(ptr(a), Hp, W("fc4.weight"), Hp, W("fc4.bias"), Bp, Sp, Hp, B, S, ptr(x), S,
                                  ptr(recon_out), S, ptr(self.dP_out), Sp, ptr(self.mse_part),
                                  ptr(self.bias_part["fc4.bias"]), st)
        # backward: decoder
        dy, kd, wname = self.dP_out, Sp, "fc4.weight"
        for i in range(d - 1, -1, -1):
            # dy [Bp,kd] is the gradient at the output of layer `wname`, whose input is dec_act[i]
            L_.rv_linear_dgrad_wgrad(ptr(dy), kd, W(wname), Hp, ptr(self.dec_act[i]), Hp, Bp, Hp, kd,
                                     ptr(self.d_dec[i]), Hp, ptr(self.bias_part["dec.%d.bias" % i]),
                                     ptr(self.slabs[wname]), Hp, self.splits[wname], *self._slab_args(wname), st)
            dy, kd, wname = self.d_dec[i], Hp, "dec.%d.weight" % i
        # dec.0: input is z (no ReLU): dz as fp32 slabs, weight gradient separately
        # (rv_latent_bwd: dz tiles with the reparameterisation backward and the loss scalar in their launch, dec.0's weight
        # gradient on the same launch's extra workgroups -- no dz slabs, no rv_reparam_bwd launch)
        L_.rv_latent_bwd(ptr(dy), Hp, W("dec.0.weight"), Lp, Bp, Hp, Lp, B, L, S, ptr(self.mulv),
                         ptr(eps if eps is not None else self.eps), self.kl_beta, None, None, ptr(self.dmulv),
                         ptr(self.dbh_part), ptr(self.mse_part), self.n_mse, ptr(self.kl_part), self.n_kl,
                         ptr(self.loss_ring), ctr, self.ring, ptr(self.z), Lp, ptr(self.slabs["dec.0.weight"]), Lp,
                         self.splits["dec.0.weight"], st)
        # backward: heads and encoder
        dy, kd, wptr, wname = self.dmulv, 2 * Lp, ptr(self.Whb), "heads.weight"
        for i in range(d - 1, -1, -1):
            L_.rv_linear_dgrad_wgrad(ptr(dy), kd, wptr, Hp, ptr(self.enc_act[i]), Hp, Bp, Hp, kd,
                                     ptr(self.d_enc[i]), Hp, ptr(self.bias_part["enc.%d.bias" % i]),
                                     ptr(self.slabs[wname]), Hp, self.splits[wname], *self._slab_args(wname), st)
            dy, kd, wname = self.d_enc[i], Hp, "enc.%d.weight" % i
            wptr = W(wname)
        L_.rv_linear_wgrad(ptr(dy), Hp, ptr(self.xb), Sp, Hp, Sp, Bp, self.splits["enc.0.weight"], self.tile_enc0,
                           ptr(self.slabs["enc.0.weight"]), Sp, *self._slab_args("enc.0.weight"), st)
        if adam:
            for chunk in self._chunks:
                L_.rv_adam_multi(chunk, len(chunk), ptr(self.param), ptr(self.exp_avg), ptr(self.exp_avg_sq), None, None,
                                 self.lr, 1.0, ctr, st)

    def gradients(self):
        """Exact-shape fp32 gradients of the last backward (sums the slabs); for tests."""
        out = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        for chunk in self._chunks:
            lib().rv_grad_finalize(chunk, len(chunk), ptr(out), 0, stream_ptr())
        return {k: self.view(out, k) for k in self.names}

    def outputs(self):
        mu = self.mulv[:self.B, :self.L].contiguous()
        lv = self.mulv[:self.B, self.Lp:self.Lp + self.L].contiguous()
        return mu, lv

    def last_loss(self):
        slot = (int(self.step_counter.item()) - 1) % self.ring
        return tuple(float(v) for v in self.loss_ring[slot, :3].tolist())

    def losses(self, n):
        done = int(self.step_counter.item())
        n = min(n, self.ring, done)
        return self.loss_ring[[(done - n + i) % self.ring for i in range(n)], 0].tolist()
