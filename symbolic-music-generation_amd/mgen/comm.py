"""Gradient exchange over RCCL through the C-ABI (`mgen_comm_*`, csrc/comm.hip): the native form of what
`torch.distributed.all_reduce` on backend "nccl" does for `GradReducer`.

Reference: lightning's DDP strategy behind `lightning.Trainer(...)` (/root/reference/src/train.py:63-72): gradient
buckets are all-reduced during backward and divided by the world size.

The 128-byte RCCL id is made by rank 0 and handed to the other ranks through `torch.distributed` (any backend; only
this one small object travels that way) or given by the caller.
"""
import ctypes

import torch
import torch.distributed as dist

from ._lib import call, lib


def rccl_version():
    v = ctypes.c_int(0)
    call("mgen_comm_rccl_version", ctypes.byref(v))
    return v.value


class NativeComm:
    def __init__(self, rank=None, world=None, device=None, unique_id=None, group=None):
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if device is None:
            device = torch.cuda.current_device()
        if unique_id is None:
            buf = ctypes.create_string_buffer(128)
            if rank == 0:
                call("mgen_comm_unique_id", buf)
            box = [buf.raw]
            if world > 1:
                src = dist.get_global_rank(group, 0) if group is not None else 0
                dist.broadcast_object_list(box, src=src, group=group)
            unique_id = box[0]
        assert len(unique_id) == 128
        self.rank, self.world, self.device = rank, world, device
        self._h = ctypes.c_void_p()
        call("mgen_comm_init", ctypes.byref(self._h), ctypes.c_char_p(unique_id), rank, world, device)
        self.last_bytes, self.last_buckets = 0, 0

    def allreduce_(self, buf, average=True):
        """In-place mean (or sum) of a contiguous fp32 device tensor over the ranks; asynchronous: ordered behind the
        current stream's work, finished for whoever calls wait()."""
        if buf.dtype != torch.float32 or not buf.is_cuda or not buf.is_contiguous():
            raise ValueError("NativeComm.allreduce_: contiguous fp32 device tensor expected")
        call("mgen_comm_allreduce_bucket", self._h, buf.data_ptr(), buf.numel(), int(average),
             torch.cuda.current_stream().cuda_stream)

    def wait(self):
        """The current stream waits for every reduction enqueued so far (no host synchronisation)."""
        b, n = ctypes.c_long(0), ctypes.c_int(0)
        call("mgen_comm_wait", self._h, torch.cuda.current_stream().cuda_stream, ctypes.byref(b), ctypes.byref(n))
        self.last_bytes, self.last_buckets = b.value, n.value

    def close(self):
        if self._h:
            lib().mgen_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
