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


class RcclUnavailable(RuntimeError):
    """The direct RCCL route could not be set up -- raised on EVERY rank of the group (the stages below are agreed
    across ranks), so the callers can fall back together."""


def _fake_failure(stage: str, rank: int) -> bool:
    """MVAE_FAKE_RCCL_INIT_FAILURE=<stage>[:<rank>] (stage: load | id | create | warmup; "1" = load; without a rank: every rank)
    makes that stage fail here -- the test hook of the agreed fall-back."""
    v = os.environ.get("MVAE_FAKE_RCCL_INIT_FAILURE", "")
    if v in ("", "0"):
        return False
    st, _, rk = v.partition(":")
    st = "load" if st == "1" else st
    return st == stage and (rk == "" or int(rk) == rank)


class FlatAllReduce:

    def __init__(self, device, group: Optional[dist.ProcessGroup] = None) -> None:
        from .distributed import agree_any
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
        self._h = C.c_void_p()

        def stage(name, fn):
            """Run one set-up stage; the ranks AGREE on its outcome (host side, through the rendezvous store) before any
            of them enters the next one, so that a rank whose librccl does not load never leaves its peers blocked in
            the collective ncclCommInitRank, and every rank raises RcclUnavailable together."""
            err = None
            try:
                if _fake_failure(name, self.rank):
                    raise RuntimeError(f"MVAE_FAKE_RCCL_INIT_FAILURE at stage {name!r}")
                fn()
            except Exception as e:  # noqa: BLE001
                err = e
            if agree_any(err is not None, group, tag="rccl-" + name):
                self.close()
                raise RcclUnavailable(f"stage {name!r} failed on " + (f"this rank ({type(err).__name__}: {err})"
                                                                       if err is not None else "another rank"))

        stage("load", lambda: check(load().mvae_rccl_load(_rccl_path())))
        ident = [None]

        def draw_id():  # rank 0 only; a failure here is agreed like any other stage BEFORE anybody enters the broadcast
            if self.rank == 0:
                buf = (C.c_uint8 * ID_BYTES)()
                check(load().mvae_rccl_unique_id(buf))
                ident[0] = bytes(buf)

        stage("id", draw_id)
        if self.world > 1:
            dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)

        def create():
            with torch.cuda.device(self.device):
                buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(ident[0])
                check(load().mvae_rccl_create(buf, self.rank, self.world, C.byref(self._h)))  # collective

        def warmup():
            with torch.cuda.device(self.device):
                # first collective outside any capture: RCCL sets up its channels / proxy threads lazily
                warm = torch.ones(64, device=self.device)
                self.all_reduce(warm)
                torch.cuda.synchronize(self.device)
                if float(warm[0].item()) != float(self.world):
                    raise RuntimeError(f"warm-up all-reduce returned {float(warm[0].item())}, expected {self.world}")

        stage("create", create)
        stage("warmup", warmup)

    def all_reduce(self, t: Tensor) -> None:
        """t <- sum over ranks of t, in place, on the current stream of the tensor's device."""
        assert t.dtype == torch.float32 and t.is_contiguous()
        check(load().mvae_flat_allreduce(self._h, ptr(t), t.numel(), stream_ptr(t.device)))

    def reduce_scatter(self, t: Tensor) -> None:
        """In place: afterwards rank r holds the sums of t[r n / world : (r + 1) n / world] in that range of t (the rest of t
        is unspecified).  t.numel() must be a multiple of the world size."""
        assert t.dtype == torch.float32 and t.is_contiguous()
        check(load().mvae_flat_reduce_scatter(self._h, ptr(t), t.numel(), stream_ptr(t.device)))

    def all_gather(self, t: Tensor) -> None:
        """In place: every rank's range t[r n / world : (r + 1) n / world] reaches everybody."""
        assert t.dtype == torch.float32 and t.is_contiguous()
        check(load().mvae_flat_allgather(self._h, ptr(t), t.numel(), stream_ptr(t.device)))

    def broadcast(self, t: Tensor, src: int = 0) -> None:
        """t of GLOBAL rank `src` (torch.distributed's convention, dist.broadcast) -> every rank of the group (float32 /
        int32 tensors, in place, current stream).  The communicator numbers its ranks inside the group."""
        assert t.element_size() == 4 and t.is_contiguous()
        root = dist.get_group_rank(self.group, int(src)) if (self.group is not None and dist.is_initialized()) else int(src)
        check(load().mvae_flat_broadcast(self._h, C.c_void_p(t.data_ptr()), t.numel(), root, stream_ptr(t.device)))

    def close(self) -> None:
        if self._h:
            load().mvae_rccl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
