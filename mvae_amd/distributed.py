"""Data-parallel step over torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

New functionality (the reference is single-device).  Samples are independent given the
parameters and the loss is a batch SUM (stats.py:200-202), so:
  * batch rows are split contiguously across ranks (`shard_rows`), every rank draws / receives its own eps rows;
  * parameters, optimizer state and radii are replicated;
  * ONE exchange per step: all-reduce(SUM) of the flat gradient buffer (P floats, 2.55 MB for h2,s2,e2) as ONE bucket
    (librccl directly on the step's stream, mvae_amd/rccl.py, by default; torch.distributed's all_reduce as the agreed
    fall-back and for engines without a HIP device), after the last backward launch; then every rank applies the identical
    optimizer step.  (Rounds 2-5 could split the exchange in two buckets -- fc_logits' half of the buffer was final one launch
    before the rest; since the four-launch step produces every weight gradient in its last launch there is nothing to overlap);
  * the epoch >= 10 gate and the radius warm-up are functions of the epoch only: no communication;
  * statistics are summed across ranks only when somebody reads them (`reduce_stats`).
`engine` is anything with `.grads` (flat tensor), `.stats`, `forward_backward(x, eps, beta)` and
`optimizer_step(do_curvature_step, batch=...)` -- the StepEngine on the GPU.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def _env_on(name: str) -> bool:
    """An environment switch is ON unless it is unset, empty or "0" (the same reading for every MVAE_* switch)."""
    import os
    return os.environ.get(name, "") not in ("", "0")


def init_from_env() -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the default process group when
    WORLD_SIZE > 1.  Backend "gloo" unless MVAE_DIST_BACKEND says otherwise: the process group is the host-side channel
    (rendezvous, the RCCL communicator id, barriers); the gradients travel on librccl directly (mvae_amd/rccl.py), so no
    ProcessGroupNCCL -- and none of its watchdog threads -- lives next to the captured steps.  MVAE_DIST_ONE_DEVICE=1
    puts every rank on cuda:0 (a flow check on a single-GPU box; the exchange then goes through gloo)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if _env_on("MVAE_DIST_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("MVAE_DIST_BACKEND", "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


_agree_seq = 0


def agree_any(flag: bool, group: Optional[dist.ProcessGroup] = None, tag: str = "") -> bool:
    """True on EVERY rank iff `flag` is true on ANY rank.  Goes through the rendezvous store (host side, no device
    work), so it is usable right after a failed stream capture, when HIP may refuse further launches.  Every rank must
    call it the same number of times in the same order.  Without a process group it returns `flag`."""
    global _agree_seq
    if not dist.is_initialized():
        return bool(flag)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return bool(flag)
    import time
    store = dist.distributed_c10d._get_default_store()
    ranks = ",".join(str(r) for r in dist.get_process_group_ranks(group)) if group is not None else "world"
    _agree_seq += 1
    key = f"mvae_amd/agree/{ranks}/{tag}/{_agree_seq}"
    store.add(key + "/flag", 1 if flag else 0)
    store.add(key + "/arrived", 1)
    deadline = time.monotonic() + 600.0
    while store.add(key + "/arrived", 0) < world:
        if time.monotonic() > deadline:
            raise RuntimeError(f"agree_any timed out waiting for the other ranks ({key}, rank {rank})")
        time.sleep(0.002)
    return store.add(key + "/flag", 0) > 0


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of the rows owned by `rank` (earlier ranks take the remainder)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DataParallelStep:

    def __init__(self, engine, group: Optional[dist.ProcessGroup] = None, always_exchange: bool = False,
                 exchange: Optional[str] = None, shard_optimizer: Optional[bool] = None) -> None:
        """always_exchange: take the gradients -> all-reduce -> optimizer route even at world size 1 (a diagnostic: it
        exercises the collective, its graph capture and k_optim on a single GPU).
        exchange (MVAE_DP_EXCHANGE): "rccl" -- ncclAllReduce on librccl DIRECTLY, enqueued on the step's own streams
        (mvae_amd/rccl.py: no ProcessGroupNCCL, no watchdog thread, captured natively; the process group is only the side
        channel for the communicator id and may be gloo) -- the default for an engine on a HIP device; "allreduce" --
        torch.distributed's all_reduce (what an engine without a HIP device gets: the gloo tests on CPU); "peer" -- the
        one-shot peer-read reduction of mvae_amd/peer.py, fused into the optimizer launch, ranks of ONE node only;
        "peer2": its two-shot form (each rank reduces 1/world of the buffer, the optimizer reads every slice from its
        owner); "peer3": the SHARDED optimizer on the same two rounds (each rank reduces its 1/world slice, applies Adam to
        it, and the ranks gather the updated PARAMETERS: no pass over the whole buffer on any rank; Adam's moments live on
        the slice's owner -- `gather_optimizer_state()`).
        shard_optimizer (MVAE_DP_SHARD_OPTIMIZER=1; routes "rccl" and "allreduce", engines with `optimizer_step_slice`): the
        all-reduce becomes reduce-scatter -> optimizer on the rank's own 1/world range -> all-gather of PARAMETERS
        (librccl: ncclReduceScatter / ncclAllGather in place on the flat buffers; torch.distributed: all_reduce + one
        broadcast per owner, gloo has no reduce-scatter).  The optimizer pass per rank shrinks with the world size; the
        price on the RCCL route is two collectives where there was one -- off by default until a node has measured it."""
        import os
        self.engine = engine
        self.group = group
        self.always_exchange = bool(always_exchange)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        on_hip = getattr(getattr(engine, "grads", None), "is_cuda", False)
        # several ranks on ONE device (the flow checks on a single-GPU box): RCCL refuses duplicate devices
        one_device = _env_on("MVAE_DIST_ONE_DEVICE") or _env_on("MVAE_BENCH_ONE_DEVICE")
        self.exchange = exchange or os.environ.get("MVAE_DP_EXCHANGE", "") or \
            ("rccl" if (on_hip and not one_device) else "allreduce")
        if self.exchange not in ("allreduce", "rccl", "peer", "peer2", "peer3"):
            raise ValueError(f"unknown gradient exchange {self.exchange!r}")
        self.peer = None
        self.rccl = None
        active = self.world > 1 or self.always_exchange
        if self.exchange in ("peer", "peer2", "peer3") and active:
            if not hasattr(engine, "_context"):
                raise ValueError("the peer-read exchange is fused into the MLP step's optimizer launch (StepEngine); "
                                 "other engines (ConvEngine) exchange through 'rccl' or 'allreduce'")
            from .peer import PeerExchange
            self.peer = PeerExchange(engine, group, two_shot=self.exchange == "peer2", sharded=self.exchange == "peer3")
        self.exchange_note = ""   # why a route other than the requested one is in use (bench.py: config.exchange)
        want_shard = _env_on("MVAE_DP_SHARD_OPTIMIZER") if shard_optimizer is None else bool(shard_optimizer)
        self.shard = bool(want_shard and active and self.exchange in ("rccl", "allreduce") and
                          hasattr(engine, "optimizer_step_slice"))
        self._fallback_group = None
        if self.exchange == "rccl" and active:
            from .rccl import FlatAllReduce, RcclUnavailable
            try:
                self.rccl = FlatAllReduce(engine.device, group)
            except RcclUnavailable as e:
                # EVERY rank is here (FlatAllReduce agrees on the outcome of each stage before going on, so a failure
                # on one rank raises RcclUnavailable on all of them): the ranks fall back TOGETHER to torch.distributed's
                # all_reduce, eager and uncaptured -- slower, but a first N > 1 run still ends with a number.
                self.rccl = None
                self.exchange = "allreduce"
                self.exchange_note = f"fallback from rccl: {e}"
                self._fallback_group = self._make_fallback_group(engine.device)
        if self.shard and self.rccl is not None:
            n = int(engine.params.numel())
            lo, hi = engine.owned_range(0, self.world)
            if n % self.world or (hi - lo) * self.world != n:  # ncclReduceScatter wants equal ranges = the optimizer's slices
                self.shard = False
                self.exchange_note += "; replicated optimizer (the flat buffer does not split evenly over the ranks)"
        self.steps_since_check = 0

    @property
    def sharded(self) -> bool:
        """Whether Adam's moments live on the slice's owner only (the sharded peer route, or shard_optimizer)."""
        return self.shard or (self.peer is not None and getattr(self.peer, "sharded", False))

    def _make_fallback_group(self, device):
        """Process group for the all_reduce fall-back when the direct RCCL route could not be set up: torch's own RCCL
        backend if it initialises and passes one collective on EVERY rank (agreed), else the existing group (gloo stages
        device tensors through the host)."""
        if self.world == 1:
            return self.group
        if dist.get_backend(self.group) == "nccl":
            return self.group
        # dist.new_group must be entered by EVERY rank of the default group: on a proper sub-group only its members are
        # here, so a new group is not attempted (the members would hang in new_group) -- the sub-group itself is used
        if self.group is not None and self.group is not dist.group.WORLD and \
                dist.get_world_size(self.group) != dist.get_world_size():
            self.exchange_note += "; all_reduce on the caller's sub-group (a new RCCL group needs every rank of the world)"
            return self.group
        ok, g = True, None
        try:
            if _env_on("MVAE_FAKE_RCCL_INIT_FAILURE") or _env_on("MVAE_DIST_ONE_DEVICE") or _env_on("MVAE_BENCH_ONE_DEVICE"):
                raise RuntimeError("torch's RCCL backend not tried (fake failure / several ranks on one device)")
            ranks = dist.get_process_group_ranks(self.group) if self.group is not None else None
            g = dist.new_group(ranks=ranks, backend="nccl")
            t = torch.ones(64, device=device)
            dist.all_reduce(t, group=g)
            torch.cuda.synchronize(device)
            ok = float(t[0].item()) == float(self.world)
        except Exception:  # noqa: BLE001
            ok = False
        if agree_any(not ok, self.group, tag="fallback-nccl"):
            self.exchange_note += "; torch RCCL backend unavailable too: all_reduce on the existing group"
            return self.group
        self.exchange_note += "; all_reduce on a torch RCCL process group"
        return g

    @property
    def capturable(self) -> bool:
        """Whether a step of this route may be captured into a HIP graph: the direct RCCL route and the peer routes
        always (their steps are kernels on the caller's streams), torch.distributed's all_reduce only on its RCCL backend."""
        if self.world == 1 and not self.always_exchange:
            return True
        if self.rccl is not None or self.peer is not None:
            return True
        if self._fallback_group is not None or self.exchange_note:
            return False  # the agreed fall-back runs eager
        return dist.is_initialized() and dist.get_backend(self.group) == "nccl"

    @property
    def _xgroup(self):
        """The group the all_reduce route reduces on (the agreed fall-back's group when there is one)."""
        return self._fallback_group if self._fallback_group is not None else self.group

    def broadcast_state(self, src: int = 0) -> None:
        """Make every rank start from rank `src`'s parameters / optimizer state."""
        if self.world > 1:
            for t in (self.engine.params, self.engine.adam_m, self.engine.adam_v, self.engine.counters):
                if self.rccl is not None:
                    self.rccl.broadcast(t, src)  # `src` is a GLOBAL rank on both routes (translated inside)
                else:
                    dist.broadcast(t, src=src, group=self._xgroup)

    def train_step(self, x_local: Tensor, eps_local: Tensor, beta: float, do_curvature_step: bool) -> None:
        eng = self.engine
        if self.world == 1 and not self.always_exchange:
            eng.train_step(x_local, eps_local, beta, do_curvature_step)
            return
        if self.peer is not None:
            eng.forward_backward(x_local, eps_local, beta)
            self.peer.publish()
            self.peer.optimizer_step(do_curvature_step, batch=x_local.shape[0])
            self.steps_since_check += 1
            return
        eng.forward_backward(x_local, eps_local, beta)
        if self.shard:
            # reduce-scatter -> optimizer on the owned range -> all-gather of parameters
            if self.rccl is not None:
                self.rccl.reduce_scatter(eng.grads)
            elif self.world > 1:  # (gloo has no reduce-scatter: the whole sum, of which the rank uses its range)
                dist.all_reduce(eng.grads, op=dist.ReduceOp.SUM, group=self._xgroup)
            eng.optimizer_step_slice(self.rank, self.world, do_curvature_step, batch=x_local.shape[0])
            if self.rccl is not None:
                self.rccl.all_gather(eng.params)
            elif self.world > 1:
                for r in range(self.world):
                    lo, hi = eng.owned_range(r, self.world)
                    if hi > lo:
                        src = dist.get_global_rank(self._xgroup, r) if self._xgroup is not None else r
                        dist.broadcast(eng.params[lo:hi], src=src, group=self._xgroup)
            return
        if self.rccl is not None:
            self.rccl.all_reduce(eng.grads)  # on the step's own stream: captured with the launches around it
        elif self.world > 1:  # (world 1 with the exchange forced and no process group: the sum of one rank is the gradient)
            dist.all_reduce(eng.grads, op=dist.ReduceOp.SUM, group=self._xgroup)
        eng.optimizer_step(do_curvature_step, batch=x_local.shape[0])

    def owned_slice(self) -> Tuple[int, int]:
        """[lo, hi) in floats of the flat buffers whose Adam moments THIS rank holds on the sharded routes ("peer3",
        shard_optimizer); the whole buffer on every other route."""
        if not self.sharded:
            return 0, int(self.engine.params.numel())
        return self.engine.owned_range(self.rank, self.world)

    def gather_optimizer_state(self) -> None:
        """Sharded routes only: make adam_m / adam_v whole on every rank (each range broadcast by its owner) -- before a
        checkpoint of the optimizer state, or before switching to another route.  A no-op elsewhere."""
        if not self.sharded or self.world == 1:
            return
        for r in range(self.world):
            lo, hi = self.engine.owned_range(r, self.world)
            if hi > lo:
                for t in (self.engine.adam_m, self.engine.adam_v):
                    if self.rccl is not None:
                        self.rccl.broadcast(t[lo:hi], dist.get_global_rank(self.group, r) if self.group is not None else r)
                    else:
                        g = self._xgroup
                        dist.broadcast(t[lo:hi], src=dist.get_global_rank(g, r) if g is not None else r, group=g)

    def reduce_stats(self) -> Tensor:
        """Global sums of the running statistics (one small all-reduce, when the host wants to log)."""
        s = self.engine.stats.clone()
        if self.world > 1:
            if self.rccl is not None:
                self.rccl.all_reduce(s)
            else:
                dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self._xgroup)
        return s

    def check_exchange(self) -> None:
        """Host-side health check of the peer routes, to be called at log / epoch boundaries (it synchronises nothing):
        a wait that timed out means a rank went on with stale or half-written gradient slots -- the ranks' parameters have
        silently diverged -- so it is an error, not a statistic."""
        if self.peer is not None and self.peer.timeouts() > 0:
            raise RuntimeError(f"peer gradient exchange: {self.peer.timeouts()} wait(s) timed out on rank {self.rank}; "
                               "the ranks are no longer synchronous (restart from the last checkpoint with "
                               "MVAE_DP_EXCHANGE=rccl, or raise the time-out)")
        self.steps_since_check = 0
