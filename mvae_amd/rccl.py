"""The flat gradient all-reduce on librccl directly (C ABI: mvae_rccl_* / mvae_flat_allreduce in include/mvae_hip.h,
csrc/mvae_rccl.hip).  New functionality -- the reference is single-device.

One communicator per process (one process per GPU), created from a unique id that rank 0 draws and the host layer hands
around through `torch.distributed` (ANY backend -- gloo will do: the process group is a side channel for 128 bytes, it
carries no gradient).  `all_reduce` enqueues ncclAllReduce on the CURRENT stream, so the exchange is ordered with the
step's launches and is captured into the step's HIP graphs like them; no ProcessGroupNCCL -- and none of its watchdog
threads, whose event queries can invalidate a capture from another thread -- exists on this route.

    dp = DataParallelStep(engine, exchange="rccl")        # the default on a HIP device when world > 1
"""
import ctypes as C
import os
from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor

from ._lib import check, load, ptr, stream_ptr

ID_BYTES = 128


def _rccl_path() -> Optional[bytes]:
    """The librccl torch itself ships (already mapped when torch.distributed's RCCL backend is built in); the system's
    copy is the library's own fallback."""
    p = os.environ.get("MVAE_RCCL_LIB")
    if p:
        return p.encode()
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand.encode() if os.path.exists(cand) else None


class FlatAllReduce:

    def __init__(self, device, group: Optional[dist.ProcessGroup] = None) -> None:
        self.device = torch.device(device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
        check(load().mvae_rccl_load(_rccl_path()))
        ident = [None]
        if self.rank == 0:
            buf = (C.c_uint8 * ID_BYTES)()
            check(load().mvae_rccl_unique_id(buf))
            ident[0] = bytes(buf)
        if self.world > 1:
            dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(ident[0])
            check(load().mvae_rccl_create(buf, self.rank, self.world, C.byref(self._h)))  # collective
            # first collective outside any capture: RCCL sets up its channels / proxy threads lazily
            warm = torch.zeros(64, device=self.device)
            self.all_reduce(warm)
            torch.cuda.synchronize(self.device)

    def all_reduce(self, t: Tensor) -> None:
        """t <- sum over ranks of t, in place, on the current stream of the tensor's device."""
        assert t.dtype == torch.float32 and t.is_contiguous()
        check(load().mvae_flat_allreduce(self._h, ptr(t), t.numel(), stream_ptr(t.device)))

    def broadcast(self, t: Tensor, src: int = 0) -> None:
        """t of rank `src` -> every rank (float32 / int32 tensors, in place, current stream)."""
        assert t.element_size() == 4 and t.is_contiguous()
        check(load().mvae_flat_broadcast(self._h, C.c_void_p(t.data_ptr()), t.numel(), int(src), stream_ptr(t.device)))

    def close(self) -> None:
        if self._h:
            load().mvae_rccl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
