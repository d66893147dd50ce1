"""Direct RCCL binding (ctypes) for the gradient all-reduce of the data-parallel step.

Why not torch.distributed's own all_reduce here: c10d runs every collective on an internal stream that first WAITS for
the caller's stream.  In this engine the caller is the weight-gradient stream, which runs milliseconds behind the main
stream -- and on MI355X a fifth queue holding long-pending cross-queue waits slows the whole 4-stream step by ~40 %
(measured on one GPU, world size 1, where the collective itself is a no-op: 18.2 -> 25.9 ms per step; the same stream
choreography WITHOUT any collective costs the same, and issued on the weight-gradient stream itself it costs nothing:
tools/rccl_phase_probe.py).  RCCL's C API takes the stream: `ncclAllReduce(..., comm, stream)` enqueues
the collective in the weight-gradient stream, in order behind the kernels that produce the slice -- no extra queue, no
cross-queue wait.  The communicator is bootstrapped over the existing torch.distributed group (the 128-byte unique id is
broadcast from rank 0); `librccl.so` is the copy PyTorch itself loaded (torch/lib), so both use one runtime.

Reference behaviour reproduced: DDP's bucketed gradient all-reduce (SUM) over all ranks,
custom_fairseq/fairseq/models/distributed_fairseq_model.py:57-67.
"""
import ctypes
import os

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128
_DTYPES = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6, torch.float32: 7,
           torch.float64: 8, torch.bfloat16: 9}       # ncclDataType_t (rccl.h)
NCCL_SUM = 0


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        L.ncclGetUniqueId.restype = ctypes.c_int
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        L.ncclCommInitRank.restype = ctypes.c_int
        L.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p]
        L.ncclAllReduce.restype = ctypes.c_int
        L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        L.ncclCommDestroy.restype = ctypes.c_int
        L.ncclGetErrorString.argtypes = [ctypes.c_int]
        L.ncclGetErrorString.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("RCCL %s failed: %s (%d)" % (what, lib().ncclGetErrorString(rc).decode(), rc))


class RcclComm:
    """one communicator over all ranks of the default torch.distributed group, for stream-ordered in-place all-reduces"""

    def __init__(self, device):
        if not dist.is_initialized():
            raise RuntimeError("RcclComm needs an initialised torch.distributed process group (bootstrap of the unique id)")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = torch.device(device)
        L = lib()
        uid = _UniqueId()
        if self.rank == 0:
            _check(L.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        box = [bytes(bytearray(uid.internal))] if self.rank == 0 else [None]
        dist.broadcast_object_list(box, src=0)
        ctypes.memmove(ctypes.byref(uid), box[0], NCCL_UNIQUE_ID_BYTES)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(L.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def all_reduce_(self, t, stream=None):
        """in-place SUM over the ranks, enqueued on `stream` (default: the current stream); returns immediately"""
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("RcclComm.all_reduce_: contiguous device tensor expected")
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        _check(lib().ncclAllReduce(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(t.numel()),
                                   _DTYPES[t.dtype], NCCL_SUM, self._comm, ctypes.c_void_p(s.cuda_stream)), "ncclAllReduce")

    def destroy(self):
        if getattr(self, "_comm", None) is not None and self._comm.value:
            lib().ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()
