"""Communicator objects over the stand-in collectives of tools/fake_collective.hip (tools/libfakecoll.so), for
rehearsals on ONE GPU -- never on the product path.

`ShmComm`: the FUNCTIONAL stand-in (`shm_allreduce`: RCCL's C signature, a real exchange between processes through shared
memory and the host).  It quacks like `ddp.RcclComm` as far as
`TrainEngine.attach_comm` / `ddp.NativeDdpRunner` / `bench.py` look: `handle`, `world`, `rank`, the three `*_addr`,
`self_test`, `destroy`.  Used by tests/ddp_shm_worker.py and by `bench.py` under RV_DDP_REHEARSAL=shm (the N > 1 branch of
the bench with the library-driven step, which RCCL itself cannot run on a one-GPU box: it refuses two ranks on a device).
"""
import ctypes as C
import os

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfakecoll.so")


def load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("tools/libfakecoll.so missing: __graft_entry__.build() compiles it")
    return C.CDLL(LIB_PATH)


class ShmComm:
    def __init__(self, lib, name, world, rank, cap):
        lib.shm_comm_create.restype = C.c_void_p
        lib.shm_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_size_t]
        lib.shm_comm_destroy.argtypes = [C.c_void_p]
        h = lib.shm_comm_create(name.encode(), world, rank, cap)
        if not h:
            raise RuntimeError("shm_comm_create failed")
        self._lib, self.handle, self.world, self.rank = lib, C.c_void_p(h), world, rank
        self.allreduce_addr = C.cast(lib.shm_allreduce, C.c_void_p)

    def set_bf16_ring(self, on):
        """bf16 payloads summed hop by hop in bf16, in ring order (world - 1 roundings per element: what a ring all-reduce
        in the payload's type does, the harshest order) instead of in fp32 with one rounding (the default, the kindest)."""
        self._lib.shm_set_bf16_ring.argtypes = [C.c_void_p, C.c_int]
        self._lib.shm_set_bf16_ring(self.handle, int(bool(on)))

    def self_test(self, device=None):
        """(RcclComm's interface; the exchange itself is exercised by the caller's first step)"""

    def destroy(self):
        if self.handle:
            self._lib.shm_comm_destroy(self.handle)
            self.handle = None
