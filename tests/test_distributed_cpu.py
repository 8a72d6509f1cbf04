"""World-size-2 `gloo` run of the data-parallel step logic on CPU.

The HIP kernels cannot run here, so the compute is a stand-in engine backed by the oracle (allowed in tests); what is
under test is mvae_amd.distributed: row sharding, ONE all-reduce(SUM) of the flat gradient buffer, identical replicated
optimizer step, lazy reduction of the statistics.  Strong-scaling parity: global batch 64 split 32/32 must give the
single-process result (the summed gradient equals the single-device gradient)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvae_amd import synthetic
from mvae_amd.distributed import DataParallelStep, shard_rows
from oracle import model as M


class OracleEngine:
    """Same surface as StepEngine (grads flat buffer, stats, forward_backward, optimizer_step), CPU oracle inside."""

    def __init__(self, spec, state):
        self.orc = M.StepOracle(spec, state)
        self.names = list(self.orc.P)
        self.sizes = [self.orc.P[n].numel() for n in self.names]
        self.grads = torch.zeros(sum(self.sizes))
        self.stats = torch.zeros(3)
        self.params = torch.cat([self.orc.P[n].detach().reshape(-1) for n in self.names])
        off = 0
        for n, k in zip(self.names, self.sizes):  # the named parameters ARE ranges of the flat buffer (as in StepEngine)
            self.orc.P[n].data = self.params[off:off + k].view_as(self.orc.P[n])
            off += k
        self.adam_m = torch.zeros(1)
        self.adam_v = torch.zeros(1)
        self.counters = torch.zeros(1)

    def forward_backward(self, x, eps, beta):
        for p in self.orc.P.values():
            p.grad = None
        out = M.forward(self.orc.spec, self.orc.P, x, eps, beta)
        (-out.elbo).backward()
        flat = [(self.orc.P[n].grad if self.orc.P[n].grad is not None else torch.zeros_like(self.orc.P[n])).reshape(-1)
                for n in self.names]
        self.grads.copy_(torch.cat(flat))
        self.stats += torch.stack([out.bce.sum(), out.kl.sum(), out.elbo]).detach()

    def optimizer_step(self, do_curv, batch=None):
        off = 0
        for n, k in zip(self.names, self.sizes):
            if self.orc.P[n].requires_grad:
                self.orc.P[n].grad = self.grads[off:off + k].view_as(self.orc.P[n]).clone()
            off += k
        self.orc.adam.step()
        if do_curv:
            for o in (self.orc.sgd_pos, self.orc.sgd_neg):
                if o is not None:
                    o.step()


    # ---- the sharded optimizer's surface (StepEngine.owned_range / optimizer_step_slice)
    def owned_range(self, rank, world):
        n = self.params.numel()
        k = (n + world - 1) // world
        return min(n, k * rank), min(n, k * (rank + 1))

    def optimizer_step_slice(self, rank, world, do_curv, batch=None):
        """The stand-in applies the whole step and then puts back everything OUTSIDE the rank's range: a rank ends with new
        parameters on its own range only, so the all-gather that follows is what makes the ranks whole (and equal)."""
        lo, hi = self.owned_range(rank, world)
        old = self.params.clone()
        self.optimizer_step(do_curv)
        self.params[:lo] = old[:lo]
        self.params[hi:] = old[hi:]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, steps, out_q, shard=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    spec = M.Spec("h2,s2,e2", in_dim=32, h_dim=16, fixed_curvature=False)
    state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0)
    xs = synthetic.binary_batches(steps, 64, 32)
    eps = synthetic.eps_batches(steps, 64, spec.total_true_dim)
    lo, hi = shard_rows(64, rank, world)
    dp = DataParallelStep(OracleEngine(spec, state0), shard_optimizer=shard)
    assert dp.shard == shard and dp.sharded == shard
    assert spec.named_shapes()[-2][0] == "fc_logits.weight"
    dp.broadcast_state()
    for s in range(steps):
        dp.train_step(xs[s, lo:hi], eps[s, lo:hi], 1.0, True)
    total = dp.reduce_stats()
    # numpy: pickled by value (torch tensors would travel as shared-memory handles that die with this process)
    out_q.put((rank, {k: v.detach().numpy().copy() for k, v in dp.engine.orc.P.items()}, total.numpy().copy()))
    dist.destroy_process_group()


def test_shard_rows():
    assert [shard_rows(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_rows(128, 7, 8) == (112, 128)
    assert shard_rows(3, 3, 4) == (3, 3)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("shard", [False, True])
def test_data_parallel_world2_matches_single_process(shard):
    """Two gloo ranks, each on half of the rows, one all-reduce of the flat gradient buffer per step: equal to the
    single-process step on the whole batch.  shard: the SHARDED optimizer route (sum -> optimizer on the rank's own range
    -> all-gather of parameters): the same result, with each rank updating its own range only."""
    steps, world = 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, q, shard)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the full batch
    spec = M.Spec("h2,s2,e2", in_dim=32, h_dim=16, fixed_curvature=False)
    orc = M.StepOracle(spec, synthetic.synthetic_state(spec.named_shapes(), radius=2.0))
    xs = synthetic.binary_batches(steps, 64, 32)
    eps = synthetic.eps_batches(steps, 64, spec.total_true_dim)
    tot = torch.zeros(3)
    for s in range(steps):
        out = orc.train_step(xs[s], eps[s], 1.0, epoch=12)
        tot += torch.stack([out.bce.sum(), out.kl.sum(), out.elbo]).detach()
    results.sort(key=lambda r: r[0])
    for n in orc.P:
        a, b = torch.from_numpy(results[0][1][n]), torch.from_numpy(results[1][1][n])
        assert torch.equal(a, b), f"ranks diverged on {n}"
        ref = orc.P[n].detach()
        assert float((a - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-7, n
    for _, _, total in results:
        assert torch.allclose(torch.from_numpy(total), tot, rtol=1e-5)
