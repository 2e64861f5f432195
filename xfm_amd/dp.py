"""Data-parallel exchange through the C-ABI's RCCL entry points (include/xfm_hip.h xfm_dp_*), for a host without torch.distributed.

The reference's exchange -- DistributedDataParallel's gradient all-reduce (accelerators/ddp_accelerator.py:34-98), the AllGather of the
contrastive features (models/xfm.py:17-50), the parameter broadcast at wrap time -- goes through torch.distributed; the product's
accelerator (accelerators/rccl_ddp_accelerator.py) keeps doing that (backend "nccl" IS RCCL on ROCm).  This module is the same three
collectives on a communicator the library owns: what a C / C++ host binds, wrapped for Python so that it can be tested here.

    id = NativeComm.unique_id()            # rank 0; 128 bytes, handed to every rank by whatever the launcher has (file, env, socket)
    comm = NativeComm(id, rank, world)     # every rank, its device current
    comm.all_reduce(arena.grad[a:b], "avg")
    comm.finalize()

Every call is enqueued on the current HIP stream.  Tensors must be contiguous fp32 / bf16 / int32 device tensors."""
import ctypes

import torch

from . import _lib
from ._lib import check

ID_BYTES = 128
_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.int32: 2}
_OPS = {"sum": 0, "avg": 1, "max": 2}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dtype(t):
    assert t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES, f"xfm_dp: contiguous fp32 / bf16 / int32 device tensor expected, got {t.dtype}"
    return _DTYPES[t.dtype]


class NativeComm:
    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(ID_BYTES)
        check(_lib.load().xfm_dp_unique_id(buf), "dp_unique_id")
        return bytes(buf.raw)

    def __init__(self, unique_id, rank, world):
        assert len(unique_id) == ID_BYTES
        self.rank, self.world = int(rank), int(world)
        self._comm = ctypes.c_void_p()
        check(_lib.load().xfm_dp_init(ctypes.c_char_p(unique_id), self.rank, self.world, ctypes.byref(self._comm)), "dp_init")

    def all_reduce(self, t, op="avg"):
        """In place; "avg" is the mean DistributedDataParallel takes."""
        check(_lib.load().xfm_dp_bucket_allreduce(self._comm, t.data_ptr(), t.numel(), _dtype(t), _OPS[op], _stream()), "dp_bucket_allreduce")
        return t

    def all_gather(self, t):
        """-> [world * t.shape[0], ...]: rank r's rows at r * t.shape[0] (models/xfm.py:17-50, the forward)."""
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        check(_lib.load().xfm_dp_allgather(self._comm, t.data_ptr(), out.data_ptr(), t.numel(), _dtype(t), _stream()), "dp_allgather")
        return out

    def broadcast(self, t, root=0):
        check(_lib.load().xfm_dp_broadcast(self._comm, t.data_ptr(), t.numel(), _dtype(t), int(root), _stream()), "dp_broadcast")
        return t

    def finalize(self):
        if self._comm:
            check(_lib.load().xfm_dp_finalize(self._comm), "dp_finalize")
            self._comm = ctypes.c_void_p()
