"""ctypes binding of librawvae_hip.so (C ABI: include/rawvae_hip.h).

The library is the product path: if it is missing or a call fails this module
raises -- there is no CPU or PyTorch fallback for GPU tensors.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librawvae_hip.so")

c_long, c_int, c_float, c_void_p = C.c_long, C.c_int, C.c_float, C.c_void_p
c_u64, c_i64 = C.c_ulonglong, C.c_longlong


class RvError(RuntimeError):
    pass


class ParamDesc(C.Structure):
    _fields_ = [("offset", c_long), ("rows", c_long), ("cols", c_long),
                ("grad_slabs", c_void_p), ("grad_ld", c_long), ("grad_split_stride", c_long),
                ("grad_splits", c_int), ("shadow_bf16", c_void_p), ("shadow_f32", c_void_p),
                ("shadow_ld", c_long), ("shadow_fp8", c_void_p), ("fp8_scale", c_void_p),
                ("grad_half", c_int), ("grad_unscale", c_void_p), ("us_ld", c_long),
                ("us_split_stride", c_long)]


class PlanBuffers(C.Structure):
    _fields_ = [("param", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("grad", c_void_p), ("workspace", c_void_p), ("step_counter", c_void_p),
                ("loss_ring", c_void_p), ("ring", c_int)]


class CommDesc(C.Structure):
    """rv_comm_desc: everything rv_plan_step_ddp needs from the caller."""
    _fields_ = [("comm", c_void_p), ("world", c_int), ("rank", c_int), ("allreduce", c_void_p),
                ("grad_bf16", c_void_p), ("comm_stream", c_void_p)]


OPT_LATENT_FUSED, OPT_FP8, OPT_SLAB_DTYPE, OPT_ROCTX, OPT_DDP_SIGNAL, OPT_DDP_W1_WIDE, OPT_DDP_WAIT_MS = 0, 1, 2, 3, 4, 5, 6
OPT_DDP_DEFER_TAIL = 9
PLAN_GEMM, PLAN_TILE, PLAN_PAIR = 0, 1, 2
TILE_AUTO, SLAB_F32, SLAB_F16 = -1, 0, 1
TILE_256x256 = 7     # RV_TILE_256x256 (include/rawvae_hip.h)
PHASE_FWD, PHASE_BWD_A, PHASE_BWD_B = 1, 2, 4
PHASE_FINALIZE_A, PHASE_ADAM, PHASE_FINALIZE_B = 8, 16, 32
PHASE_ADAM_A, PHASE_ADAM_B = 64, 128
PHASE_BWD_FC4, PHASE_BWD_CHAIN, PHASE_BWD_REST = 0x100, 0x200, 0x400
PHASE_FIN_FC4, PHASE_FIN_FC1, PHASE_FIN_MID = 0x800, 0x1000, 0x2000
PHASE_ADAM_FC4, PHASE_ADAM_FC1, PHASE_ADAM_MID = 0x4000, 0x8000, 0x10000
PHASE_ANY_ADAM = PHASE_ADAM | PHASE_ADAM_A | PHASE_ADAM_B | PHASE_ADAM_FC4 | PHASE_ADAM_FC1 | PHASE_ADAM_MID
PHASE_ALL_LOCAL = PHASE_FWD | PHASE_BWD_A | PHASE_BWD_B | PHASE_ADAM
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2   # ACT_TANH: rv_linear_fp32 only

# name -> (restype, argtypes); every int-returning entry is error-checked by _wrap.
_SIGS = {
    "rv_version": (c_int, []),
    "rv_last_error": (C.c_char_p, []),
    "rv_pad_dims": (c_int, [c_long] * 4 + [C.POINTER(c_long)] * 4),
    "rv_gemm_force_tile": (c_int, [c_int]),   # test hooks (include/rawvae_hip_diag.h), not part of the product ABI
    "rv_plan_diag_skip": (c_int, [c_void_p, C.c_uint]),
    "rv_linear_dgrad_wgrad_f32": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_long,
                                          c_void_p, c_long, c_int, c_void_p, c_long, c_int, c_void_p]),
    "rv_cast_pad_bf16": (c_int, [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                                 c_void_p, c_void_p]),
    "rv_cast_pad_fp8": (c_int, [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long, c_void_p, c_void_p]),
    "rv_reparameterize_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    "rv_tanh_bwd_pack": (c_int, [c_void_p, c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_void_p]),
    "rv_colsum_partial": (c_int, [c_void_p, c_int, c_long, c_long, c_long, c_void_p, c_long, c_void_p]),
    "rv_ew_f32": (c_int, [c_int, c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "rv_scale_by": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "rv_scale_by3": (c_int, [c_void_p, c_void_p, c_long] * 3 + [c_void_p, c_void_p]),
    "rv_linear_fwd": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_long,
                              c_int, c_void_p, c_long, c_void_p]),
    "rv_linear_fp32": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_int,
                               c_void_p, c_long, c_void_p]),
    "rv_linear_fwd_f32": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_long,
                                  c_long, c_int, c_void_p, c_long, c_void_p]),
    "rv_decode_out_loss_fwd": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_long,
                                       c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long,
                                       c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    "rv_linear_dgrad": (c_int, [c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_long,
                                c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long,
                                c_int, c_void_p]),
    "rv_linear_wgrad": (c_int, [c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_long, c_int, c_int, c_void_p, c_long,
                                   c_int, c_void_p, c_void_p]),
    "rv_linear_dgrad_wgrad": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_long,
                                         c_long, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int,
                                         c_void_p, c_void_p]),
    "rv_linear_fwd_frames": (c_int, [c_void_p, c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_long,
                                     c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p]),
    "rv_linear_wgrad_adam": (c_int, [c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_long, c_int, c_void_p, c_long,
                                     c_int, c_void_p, C.POINTER(ParamDesc), c_int, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                     c_void_p, c_int, c_void_p]),
    "rv_heads_reparam_fwd": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_long, c_long,
                                     c_int, c_void_p, c_void_p, c_void_p, c_u64, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "rv_latent_fwd": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_long, c_long,
                              c_long, c_long, c_long, c_void_p, c_void_p, c_u64, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_long, c_void_p]),
    "rv_reparam_fwd": (c_int, [c_void_p, c_int, c_long, c_long, c_long, c_long, c_void_p, c_void_p,
                               c_u64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rv_loss_fused_workspace_bytes": (c_long, []),
    "rv_loss_fused": (c_int, [c_void_p] * 4 + [c_long, c_long, c_long, c_float] + [c_void_p] * 6),
    "rv_reparameterize": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_u64, c_u64,
                                  c_void_p, c_void_p]),
    "rv_randn": (c_int, [c_void_p, c_long, c_u64, c_u64, c_void_p]),
    "rv_gather_frames": (c_int, [c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_long, c_void_p, c_void_p]),
    "rv_plan_create": (c_int, [C.POINTER(c_void_p), c_long, c_long, c_long, c_long]),
    "rv_plan_destroy": (None, [c_void_p]),
    "rv_plan_workspace_bytes": (c_long, [c_void_p]),
    "rv_plan_bind": (c_int, [c_void_p, C.POINTER(PlanBuffers)]),
    "rv_plan_refresh_shadows": (c_int, [c_void_p, c_void_p]),
    "rv_plan_descs": (c_int, [c_void_p, C.POINTER(ParamDesc), c_int]),
    "rv_plan_riders": (c_int, [c_void_p, C.POINTER(c_int), C.POINTER(c_int)]),
    "rv_plan_set_external_grads": (c_int, [c_void_p] * 6),
    "rv_plan_loss": (c_int, [c_void_p, c_float, c_void_p, c_void_p]),
    "rv_plan_set_loss_grad": (c_int, [c_void_p, c_void_p, c_void_p]),
    "rv_reparam_bwd": (c_int, [c_void_p, c_int, c_long, c_long, c_long, c_long, c_long, c_void_p,
                                   c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                   c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "rv_plan_step": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float,
                             c_float, c_int, c_u64, c_void_p]),
    "rv_plan_step_frames": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_long, c_void_p, c_long, c_long, c_void_p, c_void_p, c_float,
                                    c_float, c_float, c_int, c_u64, c_void_p]),
    "rv_params_from_flat": (c_int, [C.POINTER(ParamDesc), c_int, c_void_p, c_long, c_void_p, c_void_p]),
    "rv_plan_ddp_flush": (c_int, [c_void_p, c_void_p]),
    "rv_plan_step_ddp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_u64, c_void_p]),
    "rv_plan_buffer": (c_void_p, [c_void_p, C.c_char_p, C.POINTER(c_long)]),
    "rv_graph_begin": (c_int, [c_void_p]),
    "rv_graph_end": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "rv_graph_launch": (c_int, [c_void_p, c_void_p]),
    "rv_graph_destroy": (None, [c_void_p]),
    "rv_latent_bwd": (c_int, [c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_long, c_long, c_long, c_long, c_void_p,
                              c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                              c_void_p, c_void_p, c_int, c_void_p, c_long, c_void_p, c_long, c_int, c_void_p]),
    "rv_heads_bwd": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_long, c_void_p, c_long, c_void_p,
                             c_void_p, c_long, c_void_p]),
    "rv_gemm_plan": (c_int, [c_int, c_long, c_long, c_long, c_int] + [C.POINTER(c_int)] * 4),
    "rv_adam_multi": (c_int, [C.POINTER(ParamDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_float, c_float, c_void_p, c_void_p]),
    "rv_grad_finalize": (c_int, [C.POINTER(ParamDesc), c_int, c_void_p, c_int, c_void_p]),
    "rv_plan_set_option": (c_int, [c_void_p, c_int, c_int]),
    "rv_plan_attach_comm": (c_int, [c_void_p, C.POINTER(CommDesc)]),
}

EXPORTED = tuple(_SIGS)
_lib = None


def _wrap(fn, name):
    def call(*a):
        rc = fn(*a)
        if rc != 0:
            msg = _lib.rv_last_error()
            raise RvError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else "?"))
        return rc
    call.__name__ = name
    return call


class _Lib:
    def __init__(self, path):
        self._cdll = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(self._cdll, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
            checked = res is c_int and name != "rv_version"
            setattr(self, name, _wrap(fn, name) if checked else fn)


def lib():
    """Load librawvae_hip.so once.  Raises RvError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RvError(
                "librawvae_hip.so not found at %s: build it with `python -c \"import "
                "__graft_entry__ as g; g.build()\"` or `make -C rawaudiovae_kelsey_amd/csrc`. "
                "There is no fallback path." % LIB_PATH)
        _lib = _Lib(LIB_PATH)
    return _lib


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def stream_ptr(stream=None):
    """hipStream_t of `stream`, or of the current stream of the current device (None = the null stream).  The current
    stream is read through torch's raw accessor: `torch.cuda.current_stream()` builds a Stream object per call, ~5 us
    of host time on a path that is called several times per step."""
    if stream is not None:
        return stream.cuda_stream or None
    global _raw_stream
    if _raw_stream is None:
        import torch
        get_dev, get_raw = getattr(torch._C, "_cuda_getDevice", None), getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if get_dev is not None and get_raw is not None:
            _raw_stream = lambda: get_raw(get_dev())   # noqa: E731
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream   # noqa: E731
    return _raw_stream() or None


def _gemm_plan(what, Mp, Np, Kp, splits_in):
    o = [c_int() for _ in range(4)]   # bm, bn, splits, paired
    lib().rv_gemm_plan(what, Mp, Np, Kp, splits_in, *[C.byref(v) for v in o])
    return tuple(v.value for v in o)


def gemm_pick(Mp, Np, Kp, max_splits=16):
    """(bm, bn, splits) the library recommends for a padded Mp x Np x Kp GEMM."""
    return _gemm_plan(PLAN_GEMM, Mp, Np, Kp, max_splits)[:3]


def gemm_tile(Mp, Np, splits=1):
    """(bm, bn) block tile used for a GEMM launched with `splits` K splits."""
    return _gemm_plan(PLAN_TILE, Mp, Np, 64, splits)[:2]


def dgrad_wgrad_pick(Mp, Np, Kp):
    """(paired, bm_dgrad, splits) for the fused backward of one Linear layer."""
    bm, _, splits, paired = _gemm_plan(PLAN_PAIR, Mp, Np, Kp, 0)
    return paired, bm, splits


def pad_dims(B, S, H, L):
    o = [c_long() for _ in range(4)]
    lib().rv_pad_dims(B, S, H, L, *[C.byref(v) for v in o])
    return tuple(v.value for v in o)
