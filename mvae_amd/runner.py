"""StepRunner: drives many fused steps over HBM-resident batches, optionally as HIP graphs, optionally data-parallel.

Data parallelism (new functionality -- the reference is single-device): samples are independent given the parameters
and the loss is a batch SUM (stats.py:200-202), so each rank runs forward/backward on its own rows and ONE all-reduce
(SUM) of the flat gradient buffer over RCCL/xGMI precedes the (replicated, identical) optimizer step.
"""
from typing import List, Optional

import torch
from torch import Tensor

from .distributed import DataParallelStep, agree_any
from .engine import StepEngine


def _clear_hip_error() -> None:
    """An invalidated stream capture leaves its error code behind (hipGetLastError is read-and-clear); torch's next
    launch check would otherwise report it against an unrelated, successful launch."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        for _ in range(8):
            if hip.hipGetLastError() == 0:
                break
    except OSError:
        pass


class StepRunner:

    def __init__(self, eng: StepEngine, xs: Tensor, eps: Tensor, beta: float = 1.0, do_curvature_step: bool = False,
                 graph_steps: int = 0, world_size: int = 1, reset_every: int = 0, force_exchange: bool = False,
                 graph_plan: Optional[List[int]] = None):
        """graph_steps > 0: the resident batches are captured as graphs of `graph_steps` consecutive steps each.
        graph_plan = [n0, n1, ...] instead captures ONE graph per span of n_i
        consecutive steps (the batches cycled), so that run(n_i) issued at the start of span i is a single replay (bench.py: one graph for the
        warm-up, one for the timed region).  A plan may be longer than the resident set: batches are cycled."""
        assert xs.shape[0] == eps.shape[0] and xs.shape[0] >= 1
        self.eng, self.xs, self.eps = eng, xs, eps
        self.beta, self.do_curv = float(beta), bool(do_curvature_step)
        self.world = int(world_size)
        self.n_data = xs.shape[0]
        self.gs = int(graph_steps)
        self.plan = [int(n) for n in graph_plan] if graph_plan else None
        if self.plan is not None:
            if min(self.plan) < 1:
                raise ValueError("graph_plan entries must be >= 1")
            self.gs = max(self.plan)
        # positions of one cycle of run(); position q steps on resident batch q % n_data
        self.period = sum(self.plan) if self.plan is not None else self.n_data
        self.cursor = 0  # index of the next resident batch
        self.reset_every = int(reset_every)
        self.since_reset = 0
        self.capture_steps = 0  # steps executed while warming up / capturing (they only touch the statistics)
        self.capture_failed = False
        self.replays = 0  # graph replays issued by run()
        self.graph_steps_replayed = 0  # steps those replays held
        self._snapshot = None
        self.graphs: dict = {}  # first resident batch of a span -> (graph, number of steps)
        self.dp = DataParallelStep(eng, always_exchange=force_exchange) if (self.world > 1 or force_exchange) else None
        if self.gs > 0 and self.dp is not None and not self.dp.capturable:
            # only RCCL collectives can be captured; a host-side collective (gloo) invalidates the capture and leaves
            # the process in an unusable capture state, so it is not even attempted
            self.gs = 0
        if self.gs > 0:
            if self.plan is None and self.n_data % self.gs != 0:
                raise ValueError("the number of resident batches must be a multiple of graph_steps")
            failed = False
            try:
                self._capture()
            except Exception as e:  # noqa: BLE001
                if self.dp is None:
                    raise
                # a collective that cannot be captured on this stack must not take the run down: fall back to eager
                # launches (same arithmetic, host launch cost back on the critical path)
                import sys
                print(f"[StepRunner] graph capture with the gradient all-reduce failed ({type(e).__name__}: {e})",
                      file=sys.stderr, flush=True)
                failed = True
                _clear_hip_error()
            # A capture failure is a per-process event (a watchdog thread's event query landing inside the capture): the
            # ranks must take the SAME route afterwards, or one replays graphs / re-execs while its peers wait in an
            # eager all-reduce.  The outcome is agreed through the rendezvous store: if any rank failed, all go eager.
            if self.dp is not None and agree_any(failed, self.dp.group, "steprunner-capture"):
                self.graphs = {}
                self.gs = 0
                self.capture_failed = True  # identical on every rank; callers that can start over without graphs
                torch.cuda.synchronize()

    def _capture_mode(self) -> str:
        """(Only torch.distributed's own RCCL route needs this.)  With a ProcessGroupNCCL alive, ProcessGroupNCCL's watchdog thread polls the events of earlier collectives
        (hipEventQuery); under the default "global" capture mode such a call from ANOTHER thread is an error while this
        thread captures ("operation not permitted when stream is capturing": 2 of 12 runs died that way).
        "thread_local" restricts the check to the capturing thread."""
        return "thread_local" if self.dp is not None else "global"

    def _one(self, i: int) -> None:
        if self.dp is None:
            self.eng.train_step(self.xs[i], self.eps[i], self.beta, self.do_curv)
        else:
            self.dp.train_step(self.xs[i], self.eps[i], self.beta, self.do_curv)

    def _capture(self) -> None:
        # snapshot the state the warm-up launches below will advance, so that capturing has no net effect
        eng = self.eng
        keep = [t.clone() for t in (eng.params, eng.adam_m, eng.adam_v, eng.counters, eng.stats)]
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for i in range(min(3, self.n_data)):  # library / RCCL warm-up outside capture
                    self._one(i)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            spans = self.plan if self.plan is not None else [self.gs] * (self.n_data // self.gs)
            start = 0
            for n in spans:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=self._capture_mode()):
                    for i in range(start, start + n):
                        self._one(i % self.n_data)
                self.graphs[start] = (g, n)
                start += n
            torch.cuda.synchronize()
        finally:  # capturing (or failing to) must have no net effect on the model
            for dst, src in zip((eng.params, eng.adam_m, eng.adam_v, eng.counters, eng.stats), keep):
                dst.copy_(src)

    def _state(self):
        e = self.eng
        return (e.params, e.adam_m, e.adam_v, e.counters)

    def run(self, n_steps: int) -> None:
        """Advance exactly n_steps steps (graph replays where a whole graph fits, eager launches otherwise)."""
        if self.reset_every > 0 and self._snapshot is None:
            self._snapshot = [t.clone() for t in self._state()]
        left = int(n_steps)
        while left > 0:
            if self.reset_every > 0 and self.since_reset >= self.reset_every:
                for dst, src in zip(self._state(), self._snapshot):
                    dst.copy_(src)
                self.since_reset = 0
            span = self.graphs.get(self.cursor) if self.gs > 0 else None
            if span is not None and left >= span[1]:
                span[0].replay()
                self.replays += 1
                self.graph_steps_replayed += span[1]
                self.cursor = (self.cursor + span[1]) % self.period
                left -= span[1]
                self.since_reset += span[1]
            else:
                self._one(self.cursor % self.n_data)
                self.cursor = (self.cursor + 1) % self.period
                left -= 1
                self.since_reset += 1


class EpochRunner:
    """A training epoch with NO per-step host work: the data set lives in HBM as uint8, every step prepares its
    successor's batch (gather + dynamic binarisation + eps draw, Philox) on spare workgroups of its own launch 4
    (mvae_set_next_batch_feed; two buffer pairs in alternation, the first batch of an epoch from one mvae_prepare_batch
    launch), and the steps are replayed as HIP graphs of `graph_steps` steps.  Engines without that hook (the conv
    architecture; fold=False) run [mvae_prepare_batch, step] pairs instead -- the same bits either way.  The batch cursor
    and the Adam step counter are device-resident, so one captured graph serves every position of every epoch; only a
    change of (beta, curvature gate, trainable flags) re-captures.  Replaces the reference's DataLoader worker processes +
    per-step H2D copy + torch RNG draw (mt/data/image_reconstruction.py:44-53,70-74; vae.py:153)."""

    def __init__(self, eng, images: Tensor, batch: int, seed: int = 0, graph_steps: int = 32,
                 shuffle: bool = True, dp: Optional[DataParallelStep] = None, binarize: bool = True,
                 fold: Optional[bool] = None):
        """eng: a StepEngine (MLP) or a ConvEngine (conv architecture).  binarize: ImageDynamicBinarization (MNIST,
        image_reconstruction.py:44-53) or, False, pixel / 255 as it is (CIFAR: ToTensor only, image_reconstruction.py:123-127).
        fold: prepare batch n + 1 inside step n (default: whenever the engine can; MVAE_FEED_FOLD=0 turns it off)."""
        import ctypes as C
        import os

        from ._lib import check, load, ptr, stream_ptr
        assert images.dtype == torch.uint8 and images.dim() == 2 and images.is_cuda
        self.eng, self.images, self.B = eng, images.contiguous(), int(batch)
        self.N, self.D = images.shape
        self.nb = self.N // self.B  # full batches; a ragged tail is the caller's (eager) business
        if self.nb < 1:
            raise ValueError("data set smaller than one batch")
        self.E = eng.layout.eps_dim
        self.seed, self.shuffle = int(seed), shuffle
        self.mode = 1 if binarize else 2  # mvae_prepare_batch's `train` argument
        self.gs = max(1, min(int(graph_steps), self.nb))
        dev = images.device
        can_fold = hasattr(eng, "set_next_batch_feed")
        if fold is None:
            fold = can_fold and os.environ.get("MVAE_FEED_FOLD", "1") != "0"
        if fold and not can_fold:
            raise ValueError("this engine has no in-step input preparation")
        self.fold = bool(fold)
        # A batch size that is not a multiple of 16 (the reference CLI's default is 100) runs on the fused kernels with PADDING
        # rows: the buffers get `Bp` rows, the pipeline fills the first B of them, the rest stay zero and are masked by the
        # step (StepEngine.padded_rows / mvae_set_valid_rows).  Bp == B when the engine cannot mask.
        self.Bp = eng.padded_rows(self.B) if (hasattr(eng, "padded_rows") and dp is None) else self.B
        # two (x, eps) pairs in alternation when the step prepares its successor's batch; `par` = the pair the next step reads
        self._bufs = [(torch.zeros(self.Bp, self.D, device=dev), torch.zeros(self.Bp, self.E, device=dev))
                      for _ in range(2 if self.fold else 1)]
        self.par = 0
        self.perm = torch.arange(self.N, device=dev, dtype=torch.int32)
        self._gen = torch.Generator(device=dev).manual_seed(self.seed)
        self._graphs = {}
        self._c = (C, check, load, ptr, stream_ptr)
        # data parallel: `images` is this rank's shard, `batch` its rows of the global batch; the step becomes
        # gradients -> all-reduce -> optimizer (captured only when the backend is RCCL)
        self.dp = dp if (dp is not None and dp.world > 1) else None
        self.capturable = True
        if self.dp is not None:
            # (the direct RCCL route and the peer routes are capturable under any process-group backend)
            self.capturable = self.dp.capturable

    @property
    def x(self) -> Tensor:
        """The batch the next step reads (after a step: the one prepared for its successor)."""
        return self._bufs[self.par][0][:self.B]

    @property
    def eps(self) -> Tensor:
        return self._bufs[self.par][1][:self.B]

    def _prepare(self, train: bool = True) -> None:
        """One mvae_prepare_batch launch: the batch at the cursor into the pair the next step reads."""
        C, check, load, ptr, stream_ptr = self._c
        x, eps = self._bufs[self.par]
        check(load().mvae_prepare_batch(ptr(self.images), ptr(self.perm), self.N, self.D, self.B, self.E,
                                        C.c_uint64(self.seed), ptr(self.eng.counters), self.nb,
                                        self.mode if train else (0 if self.mode == 1 else 2),
                                        ptr(x), ptr(eps), stream_ptr(self.images.device)))

    def _pair(self, beta: float, do_curv: bool, train: bool = True) -> None:
        x, eps = self._bufs[self.par]
        if self.fold:
            nx, ne = self._bufs[self.par ^ 1]
            self.eng.set_next_batch_feed(self.Bp, self.images, self.perm, self.seed, self.nb,
                                         self.mode if train else (0 if self.mode == 1 else 2), nx, ne)
            self.par ^= 1
        else:
            self._prepare(train)
        if self.dp is None:
            self.eng.train_step(x, eps, beta, do_curv)
        else:
            self.dp.train_step(x, eps, beta, do_curv)

    def _graph(self, beta: float, do_curv: bool) -> torch.cuda.CUDAGraph:
        # the trainable flags travel in the kernel arguments, so a requires_grad toggle needs a fresh capture; so do the
        # buffer pointers: a graph is captured for the pair its first step reads
        key = (float(beta), bool(do_curv), tuple(self.eng.radius_trainable), self.eng.generation, self.par)
        if key not in self._graphs:
            # graphs of an older engine generation reference freed workspaces / a stale lr: drop them
            self._graphs = {k: v for k, v in self._graphs.items() if k[3] == self.eng.generation}
            g, failed = None, False
            eng = self.eng
            par0 = self.par
            keep = [t.clone() for t in (eng.params, eng.adam_m, eng.adam_v, eng.counters, eng.stats)]
            keep_in = [t.clone() for t in self._bufs[par0]]
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._pair(beta, do_curv)  # warm-up outside capture
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                self.par = par0
                g = torch.cuda.CUDAGraph()
                mode = "thread_local" if self.dp is not None else "global"  # see StepRunner._capture_mode
                with torch.cuda.graph(g, capture_error_mode=mode):
                    for _ in range(self.gs):
                        self._pair(beta, do_curv)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                if self.dp is None:
                    raise
                import sys
                print(f"[EpochRunner] graph capture with the gradient all-reduce failed ({type(e).__name__}: {e})",
                      file=sys.stderr, flush=True)
                failed = True
                _clear_hip_error()
            finally:  # capturing (or failing to) has no net effect on the model or on the batch waiting to be read
                for dst, src in zip((eng.params, eng.adam_m, eng.adam_v, eng.counters, eng.stats), keep):
                    dst.copy_(src)
                for dst, src in zip(self._bufs[par0], keep_in):
                    dst.copy_(src)
                self.par = par0
                if self.fold:
                    eng.set_next_batch_feed(self.Bp, None)  # a failed capture may leave the context armed
            # every rank takes the same route (see StepRunner.__init__): any failure -> all run this key eagerly
            if self.dp is not None and agree_any(failed, self.dp.group, "epochrunner-capture"):
                g = None
            self._graphs[key] = g
        return self._graphs[key]

    def run_epoch(self, beta: float, do_curv: bool, use_graphs: bool = True) -> int:
        """Runs the `nb` full batches of one epoch; returns the number of steps taken."""
        if self.shuffle:
            self.perm.copy_(torch.randperm(self.N, device=self.perm.device, generator=self._gen).to(torch.int32))
        # epoch-local position = cursor % nb: start every epoch on a multiple of nb
        cur = int(self.eng.counters[8].item())
        if cur % self.nb:
            self.eng.counters[8] = cur + (self.nb - cur % self.nb)
        if self.fold:
            self._prepare()  # the epoch's first batch (what the previous epoch's last step prepared used the old permutation)
        left = self.nb
        if use_graphs and self.capturable:
            while left >= self.gs:
                g = self._graph(beta, do_curv)  # None: this configuration could not be captured (agreed across ranks)
                if g is None:
                    break
                g.replay()
                if self.fold and (self.gs & 1):
                    self.par ^= 1
                left -= self.gs
        for _ in range(left):
            self._pair(beta, do_curv)
        return self.nb
