"""Fused training-step engine: the reference's inner loop
(`optimizer.zero_grad(); model(data); loss_function(...); loss.backward();
optimizer.step()` -- train.py:184-193, train_iterable.py:200-210) as ONE host call
that enqueues 14 hand-written gfx950 kernels through the C ABI (`rv_plan_step`).

PyTorch is used for device memory and streams only.  Parameters, Adam moments and
(optionally) gradients live in flat fp32 arenas in `PARAM_NAMES` order; the
`nn.Parameter`s of a `VAE` can be re-pointed at views of the arena so that
`state_dict()` / checkpoints keep the reference's keys and layouts.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (PHASE_ADAM, PHASE_ADAM_A, PHASE_ADAM_B, PHASE_ALL_LOCAL, PHASE_ANY_ADAM, PHASE_BWD_A,
                   PHASE_BWD_B, PHASE_BWD_CHAIN, PHASE_BWD_FC4, PHASE_BWD_REST, PHASE_FINALIZE_A, PHASE_FINALIZE_B, PHASE_FWD, PlanBuffers, lib, ptr,
                   stream_ptr)

def _ops_invalidate():
    """The API path (ops.py) caches operand shadows by tensor version; the engine writes parameters through
    the C ABI, which PyTorch's version counters do not see."""
    from . import ops
    ops.invalidate_shadows()


PARAM_NAMES = ("fc1.weight", "fc1.bias", "fc21.weight", "fc21.bias", "fc22.weight", "fc22.bias",
               "fc3.weight", "fc3.bias", "fc4.weight", "fc4.bias")


def param_shapes(S, H, L):
    return {"fc1.weight": (H, S), "fc1.bias": (H,), "fc21.weight": (L, H), "fc21.bias": (L,),
            "fc22.weight": (L, H), "fc22.bias": (L,), "fc3.weight": (H, L), "fc3.bias": (H,),
            "fc4.weight": (S, H), "fc4.bias": (S,)}


class TrainEngine:
    """Owns the arenas + workspace of one (S, H, L, B) training configuration."""

    def __init__(self, segment_length, n_units, latent_dim, batch_size, device="cuda",
                 kl_beta=1e-4, lr=1e-4, seed=0, ring=256, grad_arena=True, share=None, fp8=False,
                 slab_dtype="fp16"):
        """share: another TrainEngine of the same (S, H, L) whose parameter / Adam / gradient
        arenas, step counter and loss ring this one uses (a second batch size, e.g. the ragged
        last batch of an epoch -- DataLoader keeps it, train.py:134)."""
        self.S, self.H, self.L, self.B = int(segment_length), int(n_units), int(latent_dim), int(batch_size)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RvError("TrainEngine needs a GPU device (got %s); there is no CPU path" % device)
        self.kl_beta, self.lr, self.seed, self.ring = float(kl_beta), float(lr), int(seed), int(ring)
        L_ = lib()
        self._plan = C.c_void_p()
        L_.rv_plan_create(C.byref(self._plan), self.B, self.S, self.H, self.L)
        self.shapes = param_shapes(self.S, self.H, self.L)
        self.offsets, o = {}, 0
        for k in PARAM_NAMES:
            self.offsets[k] = o
            n = 1
            for d in self.shapes[k]:
                n *= d
            o += n
        self.n_params = o
        self.param_shape_list = [self.shapes[k] for k in PARAM_NAMES]
        self.param_sizes = [(self.offsets[PARAM_NAMES[i + 1]] if i + 1 < len(PARAM_NAMES) else o) - self.offsets[k]
                            for i, k in enumerate(PARAM_NAMES)]
        f32 = dict(dtype=torch.float32, device=self.device)
        if share is not None:
            if (share.S, share.H, share.L) != (self.S, self.H, self.L):
                raise _lib.RvError("share: engines must have the same (S, H, L)")
            self.param, self.exp_avg, self.exp_avg_sq, self.grad = share.param, share.exp_avg, share.exp_avg_sq, share.grad
            self.step_counter, self.loss_ring, self.ring = share.step_counter, share.loss_ring, share.ring
            self._shared = share._shared
        else:
            # every arena extends ARENA_SLACK elements past n_params (16-byte accesses of the last tensor's tail)
            from .ddp import ARENA_SLACK
            self._arena_full = [torch.zeros(o + ARENA_SLACK, **f32) for _ in range(4 if grad_arena else 3)]
            self.param, self.exp_avg, self.exp_avg_sq = (t[:o] for t in self._arena_full[:3])
            self.grad = self._arena_full[3][:o] if grad_arena else None
            self.step_counter = torch.zeros(1, dtype=torch.int64, device=self.device)
            self.loss_ring = torch.zeros(self.ring, 4, **f32)
            self._shared = {"version": 0, "drained": 0}   # parameter version, losses already drained
        self._shadow_version = -1
        ws_bytes = L_.rv_plan_workspace_bytes(self._plan)
        self.workspace = torch.zeros(ws_bytes + 256, dtype=torch.uint8, device=self.device)
        ws_ptr = (self.workspace.data_ptr() + 255) // 256 * 256
        self._bufs = PlanBuffers(ptr(self.param), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                 ptr(self.grad), ws_ptr, ptr(self.step_counter),
                                 ptr(self.loss_ring), self.ring)
        L_.rv_plan_bind(self._plan, C.byref(self._bufs))
        self.host_steps = 0
        # fp8 (e4m3) operands for the fc1 / fc4 forward GEMMs (BASELINE configs[4]; build extension)
        # fp8: True / "full" = forward of fc1 and fc4 AND backward of fc4 (its dgrad + wgrad pair) on e4m3 operands;
        # "fwd" = the two forward GEMMs only (round 3's path); False = bf16 everywhere
        if fp8 not in (False, True, None, "full", "fwd"):
            raise _lib.RvError("fp8=%r (expected False, True / 'full', or 'fwd')" % (fp8,))
        self.fp8 = bool(fp8)
        self.fp8_mode = 0 if not fp8 else (2 if fp8 == "fwd" else 1)
        self.fp8_x_scale = 256.0        # frames are in [-1, 1]
        self.fp8_h3_scale = 16.0        # first step only; afterwards 224 / max|h3| of the previous step
        if self.fp8:
            L_.rv_plan_set_option(self._plan, _lib.OPT_FP8, self.fp8_mode)
        # element type of the fc1 / fc4 weight-gradient split-K slabs: "fp16" (default) = block-floating-point fp16, one
        # power-of-two scale per wave tile and slab (half the bytes written and re-read, any gradient magnitude); "fp32"
        if slab_dtype not in ("fp32", "fp16"):
            raise _lib.RvError("slab_dtype %r (expected 'fp32' or 'fp16')" % (slab_dtype,))
        self.slab_dtype = slab_dtype
        L_.rv_plan_set_option(self._plan, _lib.OPT_SLAB_DTYPE, _lib.SLAB_F16 if slab_dtype == "fp16" else _lib.SLAB_F32)
        self._note_init()     # the zero fills above ran on the current stream

    # ---- stream hygiene ---------------------------------------------------
    # Arenas, workspace and shadows are (re)initialised by torch ops and library launches on whatever stream is current
    # at that moment (usually torch's default stream), while steps usually run on a stream of the caller's choice.
    # PyTorch's rule -- synchronise before using memory on another stream than the one that wrote it -- is easy to
    # miss here (`TrainEngine(...); load_params(...); step(x, stream=s)` reads the shadows on `s`), and a miss is a
    # rare, silent wrong first step (seen: two processes sharing one GPU, the other process holding the device while
    # this one's zero-fill was still queued).  So every initialising call notes an event on the stream it used and the
    # next step on ANOTHER stream waits for it once.
    def _note_init(self, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(st)
        self._shared["init_events"] = [e for e in self._shared.get("init_events", []) if e[1] != st.cuda_stream][-3:] + [(ev, st.cuda_stream)]
        self._local_init = (ev, st.cuda_stream)     # this engine's own workspace / shadows

    def _await_init(self, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        # a stream under hipGraph capture may not wait for an event from outside the capture: the host waits instead
        # (the initialisation was enqueued before the capture began)
        capturing = torch.cuda.is_current_stream_capturing()

        def wait(ev, sp):
            if sp == st.cuda_stream:
                return
            if capturing:
                ev.synchronize()
            else:
                st.wait_event(ev)
        evs = self._shared.get("init_events")
        if evs:
            for ev, sp in evs:
                wait(ev, sp)
            self._shared["init_events"] = []
        loc = getattr(self, "_local_init", None)
        if loc is not None:
            wait(*loc)
            self._local_init = None

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                lib().rv_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    # ---- parameters -----------------------------------------------------
    def view(self, arena, name):
        n = 1
        for d in self.shapes[name]:
            n *= d
        o = self.offsets[name]
        return arena[o:o + n].view(self.shapes[name])

    def param_views(self):
        return {k: self.view(self.param, k) for k in PARAM_NAMES}

    def grad_views(self):
        return {k: self.view(self.grad, k) for k in PARAM_NAMES}

    def load_params(self, params):
        """params: name -> array/tensor with the reference's nn.Linear shapes."""
        self.ddp_flush()
        with torch.no_grad():
            for k in PARAM_NAMES:
                src = params[k]
                if not torch.is_tensor(src):
                    src = torch.as_tensor(src)
                self.view(self.param, k).copy_(src.to(self.device, torch.float32))
        self.params_changed()
        self.refresh_shadows()
        self._note_init()

    def set_latent_fused(self, enable):
        """True (default): heads GEMM + reparameterisation + fc3 of the forward as one launch where the library has the
        fused kernel (padded latent width 64, hidden width a multiple of 512 up to 2048, bf16); False: always three
        launches (`rv_plan_set_option`, RV_OPT_LATENT_FUSED)."""
        lib().rv_plan_set_option(self._plan, _lib.OPT_LATENT_FUSED, int(bool(enable)))

    def set_roctx(self, enable):
        """roctx ranges (rocprofv3 --marker-trace) around the phases of every step this engine enqueues
        (`rv_plan_set_option`, RV_OPT_ROCTX); raises when no roctx library can be loaded."""
        lib().rv_plan_set_option(self._plan, _lib.OPT_ROCTX, int(bool(enable)))

    def set_fp8_scales(self, x=None, w1=None, w4=None, h3=None, freeze_h3=None, dp1=None):
        """Write entries of the fp8 state block (include/rawvae_hip.h, RV_OPT_FP8).  Weight scales are
        normally chosen by refresh_shadows (224 / max|W|); `freeze_h3` pins the activation scales -- h3's and dP1's
        (parity runs)."""
        st = self.buffer("fp8_state", torch.float32, (-1,))[:16]
        for i, v in ((0, x), (1, w1), (2, w4), (3, h3), (13, dp1)):
            if v is not None:
                st[i] = float(v)
        if freeze_h3 is not None:
            st[7] = 1.0 if freeze_h3 else 0.0

    def fp8_state(self):
        return self.buffer("fp8_state", torch.float32, (-1,))[:16].tolist()

    def refresh_shadows(self, stream=None):
        """Rebuild this engine's bf16/padded weight shadows from the fp32 arena."""
        self.ddp_flush()              # a deferred data-parallel update writes parameters and shadows: before anything else
        self._await_init(stream)      # first: the parameters may have been written on another stream, and everything
                                      # below (the fp8 maxima, the shadow rebuild) reads them on `stream`
        if self.fp8:
            with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
                st = self.buffer("fp8_state", torch.float32, (-1,))[:16]
                cur = st.tolist()
                amax1 = float(self.view(self.param, "fc1.weight").abs().max())
                amax4 = float(self.view(self.param, "fc4.weight").abs().max())
                st[0] = self.fp8_x_scale
                st[1] = 224.0 / max(amax1, 1e-12)
                st[2] = 224.0 / max(amax4, 1e-12)
                if cur[3] == 0.0:
                    st[3] = self.fp8_h3_scale
                # scale of dP4's fp8 image (the fp8 backward of fc4): |dP4| <= 2 * 2 / (B S) lands within +-224
                st[12] = 56.0 * self.B * self.S
                # first guess for dP1's image (fc1's fp8 weight gradient), of dP4's order; from the second step on it
                # follows the maximum the heads' backward measured in the step before (delayed scaling, as h3's)
                if cur[13] == 0.0:
                    st[13] = 56.0 * self.B * self.S
                st[8:10] = 0.0   # max|W| of the last update: none yet for these weights
                self.buffer("fp8_state", torch.float32, (-1,))[32:] = 0.0
        lib().rv_plan_refresh_shadows(self._plan, stream_ptr(stream))
        self._shadow_version = self._shared["version"]

    def params_changed(self):
        """Call after writing the fp32 arena from outside (load_state_dict, optimizer.step ...)."""
        self._shared["version"] += 1

    def adopt(self, module):
        """Copy a VAE module's parameters into the arena and re-point the module's
        nn.Parameters at the arena views (state_dict keys/layouts unchanged)."""
        sd = dict(module.named_parameters())
        with torch.no_grad():
            for k in PARAM_NAMES:
                v = self.view(self.param, k)
                v.copy_(sd[k].detach().to(self.device, torch.float32))
                sd[k].data = v
        self.params_changed()
        self.refresh_shadows()
        self._note_init()

    # ---- stepping -------------------------------------------------------
    def step(self, x, eps=None, recon_out=None, phases=PHASE_ALL_LOCAL, grad_scale=1.0,
             adam_from_flat=False, stream=None):
        """Enqueue the selected phases for one batch `x` [B, S] fp32 (contiguous, on device)."""
        if x.dtype != torch.float32 or not x.is_contiguous() or x.numel() != self.B * self.S:
            raise _lib.RvError("step: x must be contiguous fp32 with %d x %d elements" % (self.B, self.S))
        if eps is not None and (eps.dtype != torch.float32 or not eps.is_contiguous()
                                or eps.numel() != self.B * self.L):
            raise _lib.RvError("step: eps must be contiguous fp32 [B, L]")
        if (phases & PHASE_FWD) and self._shadow_version != self._shared["version"]:
            self.refresh_shadows(stream)   # another engine sharing the arena stepped since
        if self._shared.get("init_events") or getattr(self, "_local_init", None) is not None:
            self._await_init(stream)
        lib().rv_plan_step(self._plan, int(phases), ptr(x), ptr(eps), ptr(recon_out), self.kl_beta,
                           self.lr, float(grad_scale), int(bool(adam_from_flat)), self.seed,
                           stream_ptr(stream))
        if phases & PHASE_FWD:
            self.host_steps += 1
        if phases & PHASE_ANY_ADAM:
            self._shared["version"] += 1
            self._shadow_version = self._shared["version"]   # Adam refreshed this engine's shadows
            _ops_invalidate()

    def step_frames(self, audio, index=None, first_frame=0, eps=None, recon_out=None, phases=PHASE_ALL_LOCAL,
                    grad_scale=1.0, adam_from_flat=False, stream=None):
        """One step on B hop-strided frames of a `data.DeviceAudio` (waveform resident in HBM): `index` is an
        int64 device tensor of B frame numbers (one slice of the epoch's shuffle) or None for the B consecutive
        frames from `first_frame`.  The frames are read where the waveform lives (`rv_plan_step_frames`): fc1's operand
        from its bf16 copy by the GEMM's own tile loader, the loss target from the fp32 waveform -- no cast kernel."""
        if audio.segment_length != self.S:
            raise _lib.RvError("step_frames: dataset frames are %d samples, the engine's %d" % (audio.segment_length, self.S))
        if index is not None and (index.dtype != torch.int64 or not index.is_contiguous() or index.numel() != self.B):
            raise _lib.RvError("step_frames: index must be a contiguous int64 tensor of %d frame numbers" % self.B)
        if eps is not None and (eps.dtype != torch.float32 or not eps.is_contiguous() or eps.numel() != self.B * self.L):
            raise _lib.RvError("step_frames: eps must be contiguous fp32 [B, L]")
        if (phases & PHASE_FWD) and self._shadow_version != self._shared["version"]:
            self.refresh_shadows(stream)
        if self._shared.get("init_events") or getattr(self, "_local_init", None) is not None:
            self._await_init(stream)
        # the waveform's bf16 copy (data.DeviceAudio.audio_bf16) lets fc1's tile loader read the frames in place: no cast kernel
        a16 = getattr(audio, "audio_bf16", None)
        lib().rv_plan_step_frames(self._plan, int(phases), ptr(audio.audio), ptr(a16), audio.padded, ptr(index), int(first_frame),
                                  audio.hop_size, ptr(eps), ptr(recon_out), self.kl_beta, self.lr, float(grad_scale),
                                  int(bool(adam_from_flat)), self.seed, stream_ptr(stream))
        if phases & PHASE_FWD:
            self.host_steps += 1
        if phases & PHASE_ANY_ADAM:
            self._shared["version"] += 1
            self._shadow_version = self._shared["version"]
            _ops_invalidate()

    def attach_comm(self, comm, payload=None):
        """Data-parallel mode with the collectives issued by the library itself: `comm` is a `ddp.RcclComm`
        (RCCL communicator + the address of its all-reduce).  payload: "fp32" (the default, `ddp.DEFAULT_PAYLOAD`: the
        exact mean) or "bf16" (opt-in: half the bytes on the links; error model in DESIGN.md section 5) --
        `set_ddp_payload`."""
        d = self._comm_desc = _lib.CommDesc(comm=comm.handle, world=comm.world, rank=getattr(comm, "rank", 0),
                                            allreduce=comm.allreduce_addr, comm_stream=self._comm_stream_ptr())
        lib().rv_plan_attach_comm(self._plan, C.byref(d))
        self._comm = comm   # keep the communicator alive as long as the plan can use it
        # cross-stream edges of the all-reduce schedule: device-side flags (HIP events under stream capture; the A/B:
        # DESIGN.md section 5)
        lib().rv_plan_set_option(self._plan, _lib.OPT_DDP_SIGNAL, 1)
        # how long a flag wait behind a collective -- i.e. behind the slowest peer -- may last (default: thirty seconds)
        if os.environ.get("RV_DDP_WAIT_MS"):
            lib().rv_plan_set_option(self._plan, _lib.OPT_DDP_WAIT_MS, int(os.environ["RV_DDP_WAIT_MS"]))
        # set_ddp_w1_wide(True): fc1's weight gradient on all CUs in the all-reduce schedule (twice the local step's K
        # splits) -- faster only beside a collective whose workgroups leave room on their CUs; off by default
        self.set_ddp_w1_wide(False)
        self.ddp_payload = "fp32"
        from .ddp import DEFAULT_PAYLOAD
        self.set_ddp_payload(payload or DEFAULT_PAYLOAD)

    def set_ddp_w1_wide(self, enable):
        lib().rv_plan_set_option(self._plan, _lib.OPT_DDP_W1_WIDE, int(bool(enable)))

    def _comm_stream_ptr(self):
        st = getattr(self, "_comm_stream", None)
        return st.cuda_stream if st is not None else None

    def _reattach(self, **fields):
        """Change fields of the attached communicator descriptor (rv_comm_desc is copied by the library)."""
        self.ddp_flush()      # a deferred update belongs to the descriptor it was enqueued under
        d = getattr(self, "_comm_desc", None)
        if d is None:
            raise _lib.RvError("no communicator attached (attach_comm)")
        for k, v in fields.items():
            setattr(d, k, v)
        lib().rv_plan_attach_comm(self._plan, C.byref(d))

    @staticmethod
    def ddp_payload_default():
        from .ddp import DEFAULT_PAYLOAD
        return DEFAULT_PAYLOAD

    def set_ddp_payload(self, payload):
        """Gradient all-reduce payload of `step_ddp`: "fp32" (default; exact mean of the ranks' fp32
        gradients) or "bf16" (half the bytes: each rank's summed gradient is rounded to bf16 before the
        exchange, which then sums in bf16)."""
        if payload not in ("fp32", "bf16"):
            raise _lib.RvError("set_ddp_payload: %r (expected 'fp32' or 'bf16')" % (payload,))
        if payload == "bf16":
            if getattr(self, "_grad_bf16", None) is None:
                self._grad_bf16 = torch.zeros(self.param.numel(), dtype=torch.bfloat16, device=self.param.device)
            self._reattach(grad_bf16=self._grad_bf16.data_ptr())
        else:
            self._reattach(grad_bf16=None)
        self.ddp_payload = payload

    def step_ddp(self, x, eps=None, recon_out=None, stream=None):
        """One whole data-parallel training step in one host call (`rv_plan_step_ddp`): every rank
        calls it once per batch; gradients are averaged over ranks before Adam."""
        if x.dtype != torch.float32 or not x.is_contiguous() or x.numel() != self.B * self.S:
            raise _lib.RvError("step_ddp: x must be contiguous fp32 with %d x %d elements" % (self.B, self.S))
        if eps is not None and (eps.dtype != torch.float32 or not eps.is_contiguous()
                                or eps.numel() != self.B * self.L):
            raise _lib.RvError("step_ddp: eps must be contiguous fp32 [B, L]")
        if self._shadow_version != self._shared["version"]:
            self.refresh_shadows(stream)
        if self._shared.get("init_events") or getattr(self, "_local_init", None) is not None:
            self._await_init(stream)
        self._pick_comm_stream(stream)
        self._ddp_stream = stream
        lib().rv_plan_step_ddp(self._plan, ptr(x), ptr(eps), ptr(recon_out), self.kl_beta, self.lr, self.seed,
                               stream_ptr(stream))
        self.host_steps += 1
        self._shared["version"] += 1
        self._shadow_version = self._shared["version"]
        _ops_invalidate()

    def set_ddp_defer(self, enable):
        """True: `step_ddp` (all-reduce schedule, eager launches) leaves its last wait -- second exchange done -- and the
        update of that bucket (fc1, heads, fc3) to the NEXT `step_ddp` call, which enqueues its cast launch first: the
        compute stream has nothing else to do while the exchange is on the links (RV_OPT_DDP_DEFER_TAIL; bit-identical
        results).  Whoever reads `param` / `exp_avg*` directly, or steps ANOTHER engine that shares this arena, calls
        `ddp_flush()` first; `step`, `step_frames`, `refresh_shadows`, the state-dict and health-check methods do it
        themselves.  False (default): every step completes itself."""
        if not enable:
            self.ddp_flush()
        lib().rv_plan_set_option(self._plan, _lib.OPT_DDP_DEFER_TAIL, int(bool(enable)))

    def ddp_flush(self, stream=None):
        """Enqueue what a deferred `step_ddp` left over (no-op when nothing is pending), on that step's stream."""
        if getattr(self, "_plan", None) is None or not hasattr(self, "_ddp_stream"):
            return
        lib().rv_plan_ddp_flush(self._plan, stream_ptr(stream) if stream is not None else None)

    def _pick_comm_stream(self, stream):
        """The collectives' stream is chosen per compute stream by measurement (ddp.pick_comm_stream: two streams
        that wait on each other must not share one of the runtime's hardware queues); once per compute stream,
        never during a capture."""
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        if st.cuda_stream == getattr(self, "_comm_pick_for", None):
            return
        if torch.cuda.is_current_stream_capturing():
            return
        from . import ddp
        self._comm_stream = ddp.pick_comm_stream(st, self.device)
        if getattr(self, "_comm_desc", None) is not None:   # else attach_comm hands it over
            self._reattach(comm_stream=self._comm_stream.cuda_stream)
        self._comm_pick_for = st.cuda_stream

    def plan_descs(self, from_flat=False):
        """The plan's ten `ParamDesc`s (gradient slabs or the flat arena, shadows to refresh)."""
        arr = (_lib.ParamDesc * 10)()
        lib().rv_plan_descs(self._plan, arr, int(bool(from_flat)))
        return list(arr)

    def riders(self):
        """(first, last): tensors [first, last) of the descriptor table are updated beside fc1's weight gradient, all others
        by the step's last launch (`rv_plan_riders`)."""
        a, b = C.c_int(), C.c_int()
        lib().rv_plan_riders(self._plan, C.byref(a), C.byref(b))
        return a.value, b.value

    def buffer(self, name, dtype, shape):
        """Typed view of a workspace buffer (tests / inspection)."""
        n = C.c_long()
        p = lib().rv_plan_buffer(self._plan, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        off = p - self.workspace.data_ptr()
        raw = self.workspace[off:off + n.value]
        return raw.view(dtype).view(shape)

    def padded(self):
        return _lib.pad_dims(self.B, self.S, self.H, self.L)

    def outputs(self):
        """mu, logvar of the last forward as exact-shape fp32 copies."""
        mulv = getattr(self, "_mulv_view", None)
        if mulv is None:
            Bp, Sp, Hp, Lp = self.padded()
            mulv = self._mulv_view = self.buffer("mulv", torch.float32, (Bp, 2 * Lp))
            self._lp = Lp
        Lp = self._lp
        return mulv[:self.B, :self.L].contiguous(), mulv[:self.B, Lp:Lp + self.L].contiguous()

    def steps_done(self):
        """Number of steps started on the device (reads the device counter; synchronises)."""
        n = int(self.step_counter.item())
        self.check_ddp_signals()
        return n

    def check_ddp_signals(self):
        """The data-parallel step's device-side flag waits are bounded (5 s behind local kernels, RV_DDP_WAIT_MS --
        thirty seconds by default -- behind a collective).  One that ran out has poisoned the plan on the device: no
        optimizer update has been applied since (include/rawvae_hip.h, RV_OPT_DDP_WAIT_MS), so the parameters are those
        of the last complete step -- but the steps since are lost and this rank's peers are still exchanging.  Raises if
        any did since the engine was created.  Called wherever the host reads results back anyway; a multi-rank caller
        agrees on `ddp_timeouts()` across ranks and stops them together (train.py DataParallel.check)."""
        if getattr(self, "_comm", None) is None:
            return
        self.ddp_flush()
        fl = getattr(self, "_ddp_flags", None)
        if fl is None:
            fl = self._ddp_flags = self.buffer("ddp_flags", torch.int32, (-1,))
        n = int(fl[8].item())
        if n:
            raise _lib.RvError("data-parallel step: %d cross-stream flag wait(s) timed out -- no optimizer update has been "
                               "applied since the first one (the parameters are those of the last complete step); a peer rank "
                               "is more than RV_DDP_WAIT_MS behind or gone" % n)

    def ddp_timeouts(self):
        """Count of flag waits that ran out (0 = healthy; synchronises).  Does not raise: for callers that must first
        agree with their peer ranks on what to do."""
        if getattr(self, "_comm", None) is None:
            return 0
        self.ddp_flush()      # a deferred wait has not run yet: it counts
        fl = getattr(self, "_ddp_flags", None)
        if fl is None:
            fl = self._ddp_flags = self.buffer("ddp_flags", torch.int32, (-1,))
        return int(fl[8].item())

    def last_loss(self):
        """(total, mse, kld) of the most recent step; synchronises."""
        slot = (self.steps_done() - 1) % self.ring
        return tuple(float(v) for v in self.loss_ring[slot, :3].tolist())

    def losses(self, n):
        """Totals of the last n steps (n <= ring), oldest first; synchronises."""
        done = self.steps_done()
        n = min(n, self.ring, done)
        idx = [(done - n + i) % self.ring for i in range(n)]
        return self.loss_ring[idx, 0].tolist()


    def drain_losses(self):
        """Per-step total losses recorded since the previous drain, oldest first (one device
        sync).  Must be called at least every `ring` steps or older entries are overwritten."""
        done = self.steps_done()
        n = done - self._shared["drained"]
        if n > self.ring:
            raise _lib.RvError("drain_losses: %d steps since the last drain exceed the ring of %d" % (n, self.ring))
        idx = [(self._shared["drained"] + i) % self.ring for i in range(n)]
        self._shared["drained"] = done
        return self.loss_ring[idx, 0].tolist() if n else []

    # ---- optimizer state in torch.optim.Adam form (checkpoints, train.py:208-212) ----
    def optimizer_state_dict(self):
        self.ddp_flush()
        t = float(self.steps_done())
        state = {i: {"step": torch.tensor(t), "exp_avg": self.view(self.exp_avg, k).clone(),
                     "exp_avg_sq": self.view(self.exp_avg_sq, k).clone()} for i, k in enumerate(PARAM_NAMES)}
        group = {"lr": self.lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(PARAM_NAMES)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        self.ddp_flush()
        st = sd["state"]
        with torch.no_grad():
            for i, k in enumerate(PARAM_NAMES):
                if i in st:
                    self.view(self.exp_avg, k).copy_(st[i]["exp_avg"].to(self.device))
                    self.view(self.exp_avg_sq, k).copy_(st[i]["exp_avg_sq"].to(self.device))
            steps = [int(v["step"]) for v in st.values()] if st else [0]
            self.step_counter.fill_(max(steps))
        self._shared["drained"] = int(self.step_counter.item())
        if sd.get("param_groups"):
            self.lr = float(sd["param_groups"][0].get("lr", self.lr))
        self._note_init()


class Graph:
    """hipGraph capture of a sequence of engine calls on `stream` (a torch.cuda.Stream)."""

    def __init__(self, stream):
        self.stream = stream
        self._g = C.c_void_p()

    def __enter__(self):
        lib().rv_graph_begin(self.stream.cuda_stream)
        return self

    def __exit__(self, et, ev, tb):
        lib().rv_graph_end(self.stream.cuda_stream, C.byref(self._g))
        return False

    def launch(self, stream=None):
        lib().rv_graph_launch(self._g, (stream or self.stream).cuda_stream)

    def __del__(self):
        try:
            if self._g:
                lib().rv_graph_destroy(self._g)
                self._g = None
        except Exception:
            pass
