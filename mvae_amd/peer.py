"""One-shot peer-read gradient exchange between the ranks of one node (C ABI: mvae_peer_* in include/mvae_hip.h,
kernels in csrc/mvae_peer.hip).  New functionality -- the reference is single-device.

The intra-node alternative to the RCCL all-reduce of `DataParallelStep`: every rank publishes its flat gradient buffer
in its own HBM, the optimizer launch of every rank reads all ranks' buffers through hipIpc mappings (xGMI on a node) and
adds them in rank order -- the sums, hence the parameters, are bit-identical on every rank.  `torch.distributed` is used
ONCE, at construction, to hand the hipIpc handles and the name of the shared flag page around (any backend: gloo works).

    dp = DataParallelStep(engine, exchange="peer")        # or MVAE_DP_EXCHANGE=peer
"""
import ctypes as C
import os
import secrets
from typing import Optional

import torch
import torch.distributed as dist

from ._lib import check, load, ptr, stream_ptr

IPC_HANDLE_BYTES = 64


class PeerExchange:

    def __init__(self, engine, group: Optional[dist.ProcessGroup] = None, timeout_seconds: float = 2.0,
                 two_shot: bool = False, sharded: bool = False) -> None:
        """two_shot: every rank first reduces its own 1/world slice of all slots, the optimizer launch then reads each
        slice from its owner (2 n / world floats per link and step instead of n; one more flag round per step).
        sharded: the optimizer itself is sharded -- rank r sums slice r of every slot, applies Adam to it with ITS slice of
        the moments (radii: rank 0), leaves the new parameters in its slot, and every rank gathers the other slices from their
        owners: the two-shot form's bytes, 1 / world of the optimizer pass per rank.  Adam's m and v are then valid on the
        owner only (`DataParallelStep.gather_optimizer_state()` before they are read as a whole)."""
        self.engine = engine
        # MVAE_PEER_TIMEOUT=<seconds>: the bound of a wait for a peer's flag.  Ranks that SHARE a device (the one-device
        # rehearsals of the node layout) are time-sliced: a spinning wait holds the GPU while its peer cannot run, and the
        # 2 s default was occasionally not enough for eight processes on a loaded box.
        timeout_seconds = float(os.environ.get("MVAE_PEER_TIMEOUT", timeout_seconds))
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
        # one fresh name per job for the flag page (a stale page from a crashed run must never be picked up)
        name = [f"/mvae-{os.getpid()}-{secrets.token_hex(6)}"]
        if self.world > 1:
            dist.broadcast_object_list(name, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._h = C.c_void_p()
        with torch.cuda.device(engine.device):
            check(load().mvae_peer_create(int(engine.grads.numel()), self.world, self.rank, name[0].encode(),
                                          float(timeout_seconds), C.byref(self._h)))
            check(load().mvae_peer_set_two_shot(self._h, 2 if sharded else (1 if two_shot else 0)))
            self.sharded = bool(sharded)
            mine = (C.c_uint8 * IPC_HANDLE_BYTES)()
            check(load().mvae_peer_export(self._h, mine))
            handles = [None] * self.world
            if self.world > 1:
                dist.all_gather_object(handles, bytes(mine), group=group)
            else:
                handles[0] = bytes(mine)
            for r, hb in enumerate(handles):
                if r != self.rank:
                    buf = (C.c_uint8 * IPC_HANDLE_BYTES).from_buffer_copy(hb)
                    check(load().mvae_peer_import(self._h, r, buf))
            if self.world > 1:
                dist.barrier(group=group)  # every rank has mapped every slot before anybody publishes

    def publish(self) -> None:
        """Copy the engine's gradients into this rank's slot, raise the flag, wait for the peers (all on the stream)."""
        check(load().mvae_peer_publish(self._h, ptr(self.engine.grads), stream_ptr(self.engine.device)))

    def optimizer_step(self, do_curvature_step: bool, batch: Optional[int] = None) -> None:
        """mvae_step_optimizer with g := sum over ranks of the published slots (rank order), also written to .grads (sharded
        form: on the owner's slice only)."""
        eng = self.engine
        if batch is None:
            batch = eng._last_batch if eng._last_batch is not None else (next(iter(eng._ctx)) if eng._ctx else 1)
        check(load().mvae_step_optimizer_peer(eng._context(batch), self._h, 1 if do_curvature_step else 0,
                                              stream_ptr(eng.device)))

    def timeouts(self) -> int:
        """Number of waits this rank gave up on (0 in a healthy run; synchronises nothing)."""
        return int(load().mvae_peer_timeouts(self._h))

    def close(self) -> None:
        if self._h:
            load().mvae_peer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
