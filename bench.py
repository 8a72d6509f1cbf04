"""ELBO-steps/sec of the fused HIP step (BASELINE.json metric), with the kernel roofline and the CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): MNIST shapes, model "h2,s2,e2", learnable curvature, MLP h_dim 400, batch 128 per
GPU, float32, epoch >= 10 state (radii 2.0, curvature SGD active).  One "step" = ModelVAE.train_step: forward, ELBO,
backward, Adam + SGD on the radii (+ gradient all-reduce when N > 1).  Inputs (binarised x, eps) are synthetic
(mvae_amd/synthetic.py: MNIST-shaped stroke images, dynamically binarised) and resident in HBM before the timed
region; weights are the synthetic init.
Prints ONE JSON line on rank 0.  The top-level fields are configs[1]; at N = 1 the default invocation also times short
legs of the other single-GPU BASELINE configs and reports them under "configs": {"e6", "prod36", "conv", "epoch_pipeline", "epoch_pipeline_b100"}
(configs[0], [3], [4]); `--no-extra-configs` skips them.
"""
import argparse
import glob
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODEL = "h2,s2,e2"
B, D, H = 128, 784, 400
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: f32-input MFMA dense peak
ONE_GRAPH_MAX = int(os.environ.get("MVAE_BENCH_ONE_GRAPH_MAX", "400"))  # a timed region of up to this many steps is captured as ONE graph (one replay)


def algorithmic_per_launch(P, NH=12, Z=8, E=6):
    """Algorithmic bytes / flops of each of the six launches of one fused step at B=128: every
    tensor a launch consumes or produces counted once; a weight updated in a gradient epilogue costs 7 floats per
    element (read p, m, v; write g, p, m, v)."""
    f4 = 4.0
    return {
        "enc_fwd": dict(flops=2.0 * B * D * H, bytes=f4 * (B * D + H * D + H + B * H)),
        "latent_fwd": dict(flops=2.0 * B * H * NH + 2.0 * B * Z * H,
                           bytes=f4 * (B * H + NH * H + NH + B * E + H * Z + H + B * Z + B * NH + B * H)),
        "dec1_fwd": dict(flops=2.0 * B * H * D, bytes=f4 * (B * H + D * H + D + 2 * B * D)),
        "dec1_bwd": dict(flops=2.0 * B * H * D, bytes=f4 * (B * D + D * H + 2 * B * H + 7 * D)),
        "latent_bwd": dict(flops=2.0 * B * H * D + 2.0 * B * H * Z + 2.0 * B * NH * H,
                           bytes=f4 * (3 * B * H + H * Z + 2 * B * NH + B * E + NH * H + B * D + B * H + 7 * D * H)),
        "enc_bwd": dict(flops=2.0 * B * H * D + 2.0 * B * NH * H + 2.0 * B * H * Z,
                        bytes=f4 * (2 * B * H + B * D + B * NH + B * H + B * Z + 7 * (H * D + NH * H + H * Z + 2 * H + NH))),
    }


def cpu_baseline(seconds_budget=8.0, model=MODEL, fixed=False):
    """The oracle (CPU restatement of the reference path, pinned to golden vectors recorded from the reference) timed
    on this box's host cores: same model, same synthetic inputs, same step (fwd, ELBO, bwd, Adam + curvature SGD).
    Rows (SURVEY 8d / BASELINE.md protocol): one intra-op thread; the fastest thread count of a short probe (torch's
    default of one thread per core oversubscribes this dispatch-bound workload on a many-core host); and one float64
    row at that thread count (float64 is the reference CLI's default, mt/examples/run.py:77).  The top-level
    value / cores are the best float32 row."""
    from mvae_amd import synthetic
    from oracle import model as M
    spec = M.Spec(model, in_dim=D, h_dim=H, fixed_curvature=fixed)
    state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0)
    n_data = 16
    xs = synthetic.digits_like_batches(n_data, B)
    eps = synthetic.eps_batches(n_data, B, spec.total_true_dim)

    def run(seconds, min_steps, dtype=torch.float32):
        orc = M.StepOracle(spec, state0, dtype=dtype)
        xd, ed = xs.to(dtype), eps.to(dtype)
        for s in range(3):
            orc.train_step(xd[s % n_data], ed[s % n_data], 1.0, epoch=12)
        t0 = time.perf_counter()
        n = 0
        while True:
            orc.train_step(xd[n % n_data], ed[n % n_data], 1.0, epoch=12)
            n += 1
            if n >= min_steps and time.perf_counter() - t0 > seconds:
                break
        return n, time.perf_counter() - t0

    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    probe = {}
    for t in sorted({1, 4, 8, 16, min(32, ncpu)}):
        if t <= ncpu:
            torch.set_num_threads(t)
            n, dt = run(1.0, 5)
            probe[t] = n / dt
    best = max(probe, key=probe.get)
    rows = []
    torch.set_num_threads(1)
    n1, dt1 = run(seconds_budget * 0.3, 20)
    rows.append({"threads": 1, "dtype": "f32", "value": n1 / dt1, "steps": n1, "seconds": dt1})
    torch.set_num_threads(best)
    n, dt = run(seconds_budget * 0.5, 30)
    rows.append({"threads": best, "dtype": "f32", "value": n / dt, "steps": n, "seconds": dt})
    n64, dt64 = run(seconds_budget * 0.2, 10, dtype=torch.float64)
    rows.append({"threads": best, "dtype": "f64", "value": n64 / dt64, "steps": n64, "seconds": dt64,
                 "note": "float64 is the reference CLI's default (--doubles True, run.py:77)"})
    torch.set_num_threads(default_threads)
    return {"value": n / dt, "unit": "ELBO-steps/sec", "cores": best, "kind": "port", "rows": rows,
            "sample": f"{n} steps of the same {model} B=128 workload in {dt:.1f}s with {best} intra-op threads "
                      f"(best of a probe over {sorted(probe)} threads: "
                      f"{', '.join(f'{k}: {v:.0f}/s' for k, v in sorted(probe.items()))}; host has {ncpu} logical "
                      "cores); oracle = CPU restatement of ModelVAE.train_step, float32, Python asserts on (not -O); "
                      "rows: 1 thread f32, best-thread f32, best-thread f64"}


CONV_FLOPS_B256 = 79.5e9  # fwd 26.63 GFLOP, fwd + bwd ~79.5 GFLOP at B = 256
CONV_FWD_FLOPS_B256 = 26.63e9
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md); a split product costs SIX bf16 MACs per f32 one
CONV_MODES = {
    0: ("f32", "f32-input MFMA (v_mfma_f32_16x16x4_f32) in every contraction"),
    1: ("f32 via 3-way bf16 splits", "every LDS-tiled contraction with more than 64 output columns multiplies through exact "
        "three-way bf16 splits (six piece products on the bf16 MFMA, f32 accumulation); the small ones on the f32-input MFMA"),
    2: ("f32 (forward: f32-input MFMA; backward: exact 3-way bf16 splits on the bf16 MFMA, f32 accumulation)",
        "forward contractions (they decide the ReLU masks and the logits) on the exact f32-input MFMA, bit-identical to mode 0; "
        "backward-data and weight-gradient contractions through exact three-way bf16 splits, six piece products on the bf16 "
        "MFMA, f32 accumulation (operator-level error vs float64 <= the f32 MFMA's)"),
}


def conv_effective_peak_tf(mode):
    """The MFMA roofline of the conv step for a contraction mode: algorithmic f32 flops over the time the matrix pipes need at
    their dense peaks -- f32-input MFMA 157.3 TFLOP/s for the exact part, bf16 MFMA 2500 TFLOP/s at SIX bf16 MACs per f32
    multiply-add for the split part."""
    split = {0: 0.0, 1: CONV_FLOPS_B256, 2: CONV_FLOPS_B256 - CONV_FWD_FLOPS_B256}[mode]
    t = (CONV_FLOPS_B256 - split) / (F32_MFMA_PEAK_TF * 1e12) + 6.0 * split / (BF16_MFMA_PEAK_TF * 1e12)
    return CONV_FLOPS_B256 / t / 1e12


def conv_leg(steps, warmup, graph=True, world=1, rank=0, dist_on=False, backend=None, strong=False, force_dp=False,
             repeats=5):
    """BASELINE configs[4]: CIFAR shapes (3x32x32, soft targets), conv architecture, h_dim 8192, batch 256, model
    h2,s2,e2, learnable curvature.  One step = ConvEngine.train_step (forward, ELBO, backward, Adam + curvature SGD); the
    warm-up and the timed region are ONE HIP graph each.  MFMA-bound: the roofline is f32 MFMA flops.  N > 1 (or
    --force-dp): ConvEngine behind DataParallelStep -- forward/backward on the rank's rows, ONE all-reduce (SUM) of the flat
    gradient buffer (8.4 MB), the replicated optimizer; weak scaling = 256 rows per GPU, --strong = 256 rows in all."""
    from mvae_amd import functional as Fn, synthetic
    from mvae_amd._lib import load as _load_lib
    from mvae_amd.conv import ConvEngine
    mode = int(_load_lib().mvae_set_contraction_mode(-1))
    Bg = 256  # BASELINE's batch
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = ConvEngine([("h", 2), ("s", 2), ("e", 2)], dev, radius_trainable=[True, True, False])
    shapes = [(name, tuple(v.shape)) for name, v in eng.param_views().items()]
    eng.load_state(synthetic.synthetic_state(shapes, radius=2.0, transposed_conv=("d1", "d2", "d3")))
    n_data = 8
    if strong and world > 1:
        from mvae_amd.distributed import shard_rows
        lo, hi = shard_rows(Bg, rank, world)
        xs = synthetic.uniform_batches(n_data, Bg, 3072)[:, lo:hi].contiguous().to(dev)
        eps = synthetic.eps_batches(n_data, Bg, 6)[:, lo:hi].contiguous().to(dev)
    else:
        xs = synthetic.uniform_batches(n_data, Bg, 3072, seed=4321 + rank).to(dev)
        eps = synthetic.eps_batches(n_data, Bg, 6, rank=rank).to(dev)
    Bc = xs.shape[1]
    dp = None
    if dist_on:
        import torch.distributed as dist
        from mvae_amd.distributed import DataParallelStep
        dp = DataParallelStep(eng, always_exchange=force_dp)
        dp.broadcast_state()
    one_step = eng.train_step if dp is None else dp.train_step
    graph = graph and (dp is None or dp.capturable)

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(3):
        one_step(xs[i % n_data], eps[i % n_data], 1.0, True)
    torch.cuda.synchronize()
    graphs = {}
    if graph:
        keep = [t.clone() for t in (eng.params, eng.adam_m, eng.adam_v, eng.counters, eng.stats)]
        for n in sorted({warmup, steps} - {0}):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(n):
                    one_step(xs[i % n_data], eps[i % n_data], 1.0, True)
            graphs[n] = g
        torch.cuda.synchronize()
        for dst, src in zip((eng.params, eng.adam_m, eng.adam_v, eng.counters, eng.stats), keep):
            dst.copy_(src)

    def run(n):
        if n in graphs:
            graphs[n].replay()
        else:
            for i in range(n):
                one_step(xs[i % n_data], eps[i % n_data], 1.0, True)

    run(warmup)
    sync_all()
    times = []
    for _ in range(repeats):  # every repeat: EXACTLY `steps` steps between barrier + synchronize, max over ranks
        t0 = time.perf_counter()
        run(steps)
        sync_all()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        times.append(dt)
    dt = sorted(times)[len(times) // 2]  # the median repeat
    st = eng.read_stats()["last"]
    assert st["elbo"] == st["elbo"], "non-finite ELBO"
    if rank != 0:
        return None
    step_s = dt / steps
    scale = 1 if (strong and world > 1) else world  # units (256-row steps) all ranks processed per step of the job
    flops_step = CONV_FLOPS_B256 * (Bc * world / Bg)
    # the dominant kernel, timed live with events on the launch stream: the e2 forward contraction as the step launches it
    # (implicit form: [B*16, 16*128] patches gathered by the LDS-DMA requests of k_gemm_f32pp x [512, 2048]^T + bias + ReLU,
    # 2.15 GFLOP at B = 256), the largest single launch of the step
    from mvae_amd import conv as C
    a1 = torch.randn(Bc * 64, 128, device=dev)   # NHWC activations of e1: [B, 8, 8, 128]
    w = torch.randn(512, 2048, device=dev) * 0.02  # taps-major weight as the engine holds it
    b = torch.zeros(512, device=dev)
    for _ in range(5):
        C._conv_e2(a1, w, b, Bc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        C._conv_e2(a1, w, b, Bc)
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / 20
    k_tf = 2.0 * Bc * 16 * 2048 * 512 / (k_ms * 1e-3) / 1e12
    traffic, traffic_src = conv_kernel_traffic("e2f") if Bc == 256 else (None, "none: recorded at 256 rows only")
    tf = flops_step / step_s / 1e12 / world  # per GPU
    return {
        "metric": "ELBO-steps/sec (batch 256) CIFAR conv h2,s2,e2",
        "value": steps * scale / dt, "unit": "ELBO-steps/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "strong" if (strong and world > 1) else "weak",
        "vs_baseline": None, "dtype": CONV_MODES[mode][0], "data": "synthetic",
        "config": {"contraction_mode": mode, "contraction_mode_meaning": CONV_MODES[mode][1],
                   "workload": f"BASELINE configs[4] on {world} GPU(s): CIFAR shapes (3x32x32, U[0,1] soft targets), model "
                               "h2,s2,e2, learnable curvature, conv architecture h_dim=8192, " +
                               ("global batch 256 split by rows" if (strong and world > 1) else "batch 256 per GPU") +
                               ", epoch>=10 state",
                   "global_batch": Bc * world, "parallelism": f"dp{world}" + ("(forced exchange)" if force_dp else ""),
                   "exchange": (dp.exchange + (f" ({dp.exchange_note})" if dp.exchange_note else "")) if dp is not None else "none",
                   "graph_steps": steps if graphs else 0, "graph_replays": 1 if graphs else 0,
                   "timed_repeats": repeats,
                   "repeat_ms_per_step": {"median": step_s * 1e3, "first": times[0] / steps * 1e3,
                                          "min": min(times) / steps * 1e3, "max": max(times) / steps * 1e3},
                   "final_elbo_per_sample": st["elbo"] / Bc},
        "roofline": {"bound": "mfma", "achieved": tf, "peak": conv_effective_peak_tf(mode), "unit": "TFLOP/s",
                     "frac": tf / conv_effective_peak_tf(mode),
                     "frac_of_f32_mfma_peak": tf / F32_MFMA_PEAK_TF,
                     "scope": "whole step, per GPU: SURVEY 8(d) 79.5 GFLOP (fwd + bwd contractions at 256 rows) over ms_per_step; "
                              "peak = those flops over the time the matrix pipes need at their dense peaks in this contraction "
                              "mode (f32-input MFMA 157.3 TF for the exact part; bf16 MFMA 2500 TF at six bf16 MACs per f32 "
                              "multiply-add for the split part)",
                     "kernel": "k_gemm_f32pp (gathered A operand): e2 forward contraction [4096 x 2048] x [512 x 2048]^T + bias "
                               "+ ReLU, exact f32 MFMA in every contraction mode",
                     "kernel_ms": k_ms, "kernel_achieved_TFLOPs": k_tf, "kernel_mfma_frac": k_tf / F32_MFMA_PEAK_TF,
                     "kernel_algorithmic_bytes": Bc * 64 * 128 * 4 + 512 * 2048 * 4 + Bc * 16 * 512 * 4,
                     "traffic": traffic, "traffic_source": traffic_src},
    }


def conv_kernel_traffic(op):
    """(bytes per launch, source) of one conv contraction from the newest profiles/r*_conv_pmc_traffic.json recorded from
    THIS build of the kernels (tools/pmc_conv_traffic.sh: FETCH_SIZE / WRITE_SIZE in their own passes, one op per process;
    FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note); (None, why) when there is none."""
    import glob
    from mvae_amd.build import source_hash
    cur, stale = source_hash(), None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_conv_pmc_traffic.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        note = ""
        if doc.get("source_hash") != cur:
            # another revision of the sources: still valid for the conv kernels when THEIR translation units and every header
            # are byte-identical (mvae_amd.build.conv_file_hashes, recorded in the summary as `conv_file_hashes`)
            from mvae_amd.build import conv_file_hashes
            rec = doc.get("conv_file_hashes")
            if not rec or rec != conv_file_hashes():
                stale = stale or os.path.basename(path)
                continue
            note = f" (collected from source_hash {doc.get('source_hash')}; the conv translation units and all headers are unchanged)"
        k = doc.get("kernels", {}).get(op)
        if k and "traffic_bytes" in k:
            return int(k["traffic_bytes"]), "profiles/" + os.path.basename(path) + note
    return None, (f"none: profiles/{stale} was recorded from another build of the kernels (source_hash mismatch)" if stale
                  else "none: no profiles/r*_conv_pmc_traffic.json")



def init_dist(force_dp):
    """(world, rank, local_rank, dist_on, backend): the torchrun environment; initialises the process group when the run
    is data-parallel (or the exchange is forced at world size 1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or force_dp
    backend = None
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # The process group is the HOST-side channel only (rendezvous, the RCCL communicator id, barriers): gloo.  The
        # gradients are all-reduced by librccl directly on the step's streams (mvae_amd/rccl.py), captured into the
        # graphs; no ProcessGroupNCCL / watchdog thread exists.  MVAE_BENCH_BACKEND=nccl + MVAE_DP_EXCHANGE=allreduce
        # selects torch.distributed's own RCCL route instead.  MVAE_BENCH_ONE_DEVICE=1: a dry run of the N > 1 flow on
        # a single-GPU box (all ranks on cuda:0, the exchange through gloo or the peer routes).  Not a measurement.
        backend = os.environ.get("MVAE_BENCH_BACKEND", "gloo")
        if os.environ.get("MVAE_BENCH_ONE_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(torch.device("cuda", local_rank))
    return world, rank, local_rank, dist_on, backend


def parse_model(model):
    from mvae_amd.utils import parse_component_str
    comps = []
    for tok in model.lower().split(","):
        mult, letter, dim = parse_component_str(tok.strip())
        comps += [(letter, dim)] * mult
    return comps


def make_plan(steps, warmup, graph_steps):
    """How run(warmup); run(steps) is issued: (graph_steps, graph_plan).  Up to ONE_GRAPH_MAX steps the timed region is
    ONE graph (and the warm-up another): a single replay, no per-replay launch cost inside the timed region.  Longer
    runs replay `graph_steps`-step graphs (the cost of a replay, ~10-16 us, is then < 1 % of a graph's duration)."""
    if graph_steps <= 0:
        return 0, None
    if steps <= ONE_GRAPH_MAX and warmup <= ONE_GRAPH_MAX:
        return max(steps, warmup, 1), ([warmup, steps] if warmup > 0 else [steps])
    if steps % graph_steps or warmup % graph_steps:
        g = math.gcd(steps, warmup) if warmup > 0 else steps
        graph_steps = max(d for d in range(1, min(g, graph_steps) + 1) if g % d == 0)
    return graph_steps, None


def mlp_roofline(eng, prof, step_s, fixed, pmc_config=None):
    """Whole-step roofline + per-launch table."""
    alg = algorithmic_per_launch(eng.flat.n_logical_params(), eng.layout.heads_dim, eng.layout.z_dim,
                                 eng.layout.eps_dim)
    prof = dict(prof)
    if prof.get("latent_fwd", 1.0) == 0.0:  # the fused forward: launches 2 + 3 are ONE launch (k_fwd23); hd stays in LDS
        f4, NHh, Zz = 4.0, eng.layout.heads_dim, eng.layout.z_dim
        alg["latent_dec1_fwd"] = dict(
            flops=alg["latent_fwd"]["flops"] + alg["dec1_fwd"]["flops"],
            bytes=f4 * (B * H + NHh * H + NHh + B * eng.layout.eps_dim + H * Zz + H + D * H + D + 2 * B * D + B * H + B * Zz))
        prof["latent_dec1_fwd"] = prof.pop("dec1_fwd")
        del prof["latent_fwd"], alg["latent_fwd"], alg["dec1_fwd"]
        prof = {k: prof[k] for k in ("enc_fwd", "latent_dec1_fwd", "dec1_bwd", "latent_bwd", "enc_bwd")}
    if prof.get("latent_bwd", 1.0) == 0.0:  # the four-launch step: launches 5 + 6 are ONE launch (k_bwd56)
        alg["latent_enc_bwd"] = dict(flops=alg["latent_bwd"]["flops"] + alg["enc_bwd"]["flops"],
                                     bytes=alg["latent_bwd"]["bytes"] + alg["enc_bwd"]["bytes"])
        prof["latent_enc_bwd"] = prof.pop("enc_bwd")
        del prof["latent_bwd"], alg["latent_bwd"], alg["enc_bwd"]
    # The headline fraction is the WHOLE STEP against the roofline: SURVEY section 8(d)'s algorithmic bytes / flops of one
    # step (each tensor once: x, eps, and p, m, v, g read + written) over the measured ms_per_step.  `kernel` names the
    # longest launch -- whatever it is -- with its own numbers; every launch is listed in `per_kernel`.
    P_log = eng.flat.n_logical_params()
    step_bytes_8d = 4.0 * (B * D + B * eng.layout.eps_dim + 8 * P_log)
    step_flops = sum(v["flops"] for v in alg.values())
    step_gbs, step_tf = step_bytes_8d / step_s / 1e9, step_flops / step_s / 1e12
    dom = max(prof, key=prof.get)
    dur_s = prof[dom] * 1e-3
    per_kernel = {k: {"ms": prof[k], "bytes": alg[k]["bytes"], "flops": alg[k]["flops"],
                      "hbm_frac": alg[k]["bytes"] / (prof[k] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "mfma_frac": alg[k]["flops"] / (prof[k] * 1e-3) / 1e12 / F32_MFMA_PEAK_TF} for k in prof}
    traffic, traffic_step, traffic_src = None, None, None
    # fabric bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs, see profiles/README.md): the
    # newest summary recorded from THIS build of the step kernels (its `source_hash` equals the hash of the sources the
    # loaded library was built from); a summary of an older build is stale the moment a kernel changes and is refused
    from mvae_amd.build import source_hash
    cur = source_hash()

    def _order(path):  # r02 (the round's final set) after r02a, r02b (mid-round sets) after r01
        tag = os.path.basename(path).split("_")[0]
        digits = "".join(ch for ch in tag[1:] if ch.isdigit())
        suffix = tag[1 + len(digits):]
        return (int(digits or 0), 1 if suffix == "" else 0, suffix)
    stale = None
    cur_isa = None  # machine-code fingerprints of this build's step kernels (computed only if a summary of another revision is met)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), key=_order, reverse=True):
        name = os.path.basename(path)
        if "_conv_" in name:  # (the conv engine's summaries: conv_kernel_traffic)
            continue
        try:
            with open(path) as fh:
                doc = json.load(fh)
            kern = doc.get("configs", {}).get(pmc_config) if pmc_config else doc["kernels"]
            same_code = None
            if doc.get("source_hash") != cur:
                # another revision of the sources: its counters still describe a launch whose MACHINE CODE is identical in this
                # build (mvae_amd.build.kernel_isa: fingerprints of the disassembly, recorded in the summary next to source_hash)
                rec = (doc.get("kernel_isa") or {}).get("hashes")
                if rec and kern:
                    if cur_isa is None:
                        from mvae_amd.build import kernel_isa
                        cur_isa = kernel_isa((doc.get("kernel_isa") or {}).get("unit", "mvae_step")) or {}
                    same_code = {k for k in kern if rec.get(k) is not None and rec.get(k) == cur_isa.get(k)}
                if not same_code:
                    stale = stale or name
                    continue
                kern = {k: v for k, v in kern.items() if k in same_code}  # (a launch whose code changed raises KeyError below)
            if kern is None:
                continue
            # a profile slot -> the kernels that can fill it (the fused / block / per-row / wave-cooperative paths)
            cands = {"enc_fwd": [["k_enc_fwd"]], "latent_dec1_fwd": [["k_fwd23"]],
                     "latent_fwd": [["k_heads_comp"], ["k_latent_fwd"]],
                     "dec1_fwd": [["k_fwd3m"], ["k_dec1_fwd_duals"], ["k_dec1_fwd"]], "dec1_bwd": [["k_dec1_bwd"]],
                     "latent_bwd": [["k_latent_bwd_blk"], ["k_latent_bwd"]],
                     "enc_bwd": [["k_enc_bwd3"], ["k_enc_bwd"]],
                     "latent_enc_bwd": [["k_bwd56"]]}

            def slot_bytes(slot):
                for group in cands[slot]:
                    if all(g in kern for g in group):
                        return sum(kern[g]["traffic_bytes"] for g in group)
                raise KeyError(slot)
            try:
                t_dom, t_step = slot_bytes(dom), sum(slot_bytes(k) for k in prof)
            except KeyError:
                if same_code is not None:  # a launch of this step has other machine code than the summary's revision
                    stale = stale or name
                raise
            traffic, traffic_step = t_dom, t_step
            traffic_src = "profiles/" + name + (f" [configs.{pmc_config}]" if pmc_config else "")
            if same_code is not None:
                traffic_src += (f" (collected from source_hash {doc.get('source_hash')}; the machine code of this step's "
                                f"{len(prof)} launches is identical in this build: build.kernel_isa)")
            break
        except (OSError, KeyError, ValueError):
            continue
    if traffic is None and stale:
        traffic_src = f"none: profiles/{stale} was recorded from another build of the kernels (source_hash mismatch)"
    if step_gbs / HBM_PEAK_GBS >= step_tf / F32_MFMA_PEAK_TF:
        roof = {"bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": step_gbs / HBM_PEAK_GBS}
    else:
        roof = {"bound": "mfma", "achieved": step_tf, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": step_tf / F32_MFMA_PEAK_TF}
    roof.update({
        "scope": "whole step: SURVEY 8(d) bytes 4(BD + B*eps_dim + 8P) and GEMM flops over ms_per_step",
        "step_bytes": step_bytes_8d, "step_flops": step_flops,
        "step_hbm_frac": step_gbs / HBM_PEAK_GBS, "step_mfma_frac": step_tf / F32_MFMA_PEAK_TF,
        "kernel": dom, "kernel_rule": "the longest launch",
        "kernel_achieved_GBps": alg[dom]["bytes"] / dur_s / 1e9, "kernel_hbm_frac": per_kernel[dom]["hbm_frac"],
        "kernel_achieved_TFLOPs": alg[dom]["flops"] / dur_s / 1e12, "kernel_mfma_frac": per_kernel[dom]["mfma_frac"],
        "traffic": traffic, "traffic_step": traffic_step, "traffic_source": traffic_src,
        "launch_bytes_sum": sum(v["bytes"] for v in alg.values()),
        "kernel_ms": prof, "per_kernel": per_kernel})
    return roof


def mlp_leg(model, fixed, steps, warmup, dev, repeats=5):
    """A short single-GPU leg of another BASELINE MLP config (one graph for the warm-up, one for the timed region)."""
    from mvae_amd import synthetic
    from mvae_amd.engine import StepEngine
    from mvae_amd.runner import StepRunner
    comps = parse_model(model)
    eng = StepEngine(comps, D, H, dev, radius_trainable=[not fixed] * len(comps), lr=1e-3)
    eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
    n_data = 64
    xs = synthetic.digits_like_batches(n_data, B, seed=4321).to(dev)
    eps = synthetic.eps_batches(n_data, B, eng.layout.eps_dim, rank=0).to(dev)
    gs, plan = make_plan(steps, warmup, 50)
    runner = StepRunner(eng, xs, eps, beta=1.0, do_curvature_step=not fixed, graph_steps=gs, graph_plan=plan)
    runner.run(warmup)
    torch.cuda.synchronize()
    times = []
    for _ in range(repeats):
        if plan is not None and runner.gs > 0 and warmup > 0:
            runner.cursor = warmup  # every repeat replays the timed graph
        t0 = time.perf_counter()
        runner.run(steps)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]  # the median repeat
    stats = eng.read_stats()
    assert stats["sum"]["steps"] == warmup + repeats * steps, stats["sum"]["steps"]
    assert stats["last"]["elbo"] == stats["last"]["elbo"], "non-finite ELBO"
    prof = eng.profile_step(xs[0], eps[0], 1.0, not fixed, iters=100)
    roof = mlp_roofline(eng, prof, dt / steps, fixed, pmc_config={"e6": "e6", "6h2,6s2,6e2": "prod36", "h40": "h40"}.get(model))
    return {"metric": f"ELBO-steps/sec (batch 128) MNIST {model}", "value": steps / dt, "unit": "ELBO-steps/sec",
            "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "dtype": "f32",
            "workload": f"MNIST shapes (D=784), model {model}, {'fixed' if fixed else 'learnable'} curvature, "
                        "MLP h_dim=400, batch 128, epoch>=10 state",
            "graph_replays": runner.replays, "timed_repeats": repeats,
            "repeat_ms_per_step": {"median": dt / steps * 1e3, "first": times[0] / steps * 1e3,
                                   "min": min(times) / steps * 1e3, "max": max(times) / steps * 1e3},
            "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms",
                                              "step_hbm_frac", "step_mfma_frac", "traffic", "traffic_step", "traffic_source")}}


def epoch_pipeline_leg(dev, epochs=3, batch=None):
    """Scope row f-2 measured where the driver looks: whole training epochs of BASELINE configs[1] through the device-side
    input pipeline -- a 60000 x 784 uint8 synthetic set resident in HBM, batches gathered by a device permutation, dynamic
    binarisation and the eps draw (Philox) done by spare workgroups of the PREVIOUS step (mvae_set_next_batch_feed), steps
    replayed as HIP graphs.  Unlike the headline (inputs already prepared in HBM), every timed step here includes preparing its
    successor's inputs; the per-epoch permutation and one mvae_prepare_batch launch per epoch are inside the timed region."""
    from mvae_amd import synthetic
    from mvae_amd.engine import StepEngine
    from mvae_amd.runner import EpochRunner
    comps = parse_model(MODEL)
    eng = StepEngine(comps, D, H, dev, radius_trainable=[True] * len(comps), lr=1e-3)
    eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
    images = (torch.rand(60000, D, device=dev) ** 3 * 255).to(torch.uint8)
    batch = B if batch is None else int(batch)  # (100: the reference CLI's default, mt/examples/run.py:32 -- buffers padded to 112 rows)
    er = EpochRunner(eng, images, batch, seed=1)
    for _ in range(2):
        n = er.run_epoch(1.0, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        n = er.run_epoch(1.0, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = eng.read_stats()
    assert stats["sum"]["steps"] == (2 + epochs) * n, stats["sum"]["steps"]  # (capture warm-ups restore the statistics)
    assert stats["last"]["elbo"] == stats["last"]["elbo"], "non-finite ELBO"
    return {"metric": f"ELBO-steps/sec (batch {batch}) MNIST {MODEL}, whole epochs through the device-side input pipeline",
            "batch": batch, "buffer_rows": er.Bp, "kernel_path": eng.kernel_path(er.Bp),
            "value": n * epochs / dt, "unit": "ELBO-steps/sec", "ms_per_step": dt / (n * epochs) * 1e3,
            "steps": n * epochs, "epochs": epochs, "steps_per_epoch": n, "dtype": "f32",
            "workload": "60000 x 784 uint8 synthetic images in HBM; per step: gather by a device permutation + dynamic "
                        "binarisation + eps draw (Philox4x32-10) for the NEXT step inside launch 4, fused train step; "
                        f"HIP graphs of {er.gs} steps; in_step_preparation={er.fold}",
            "baseline_config": "configs[1] with the reference's DataLoader + ImageDynamicBinarization replaced (SURVEY 8 f-2)"}


def loglik_leg(dev, n=500, iters=40):
    """Scope row f-1 (ModelVAE.log_likelihood, vae.py:82-123) at the reference's evaluation setting: B = 128, n = 500
    importance samples, model h2,s2,e2, MLP h_dim 400 -- 64 000 decoded rows per batch.  MFMA-bound: the decoder's two
    layers are 2 * 64000 * (Z*H + H*D) flops.  A timed repeat is `iters` back-to-back calls between two synchronisations (the
    reference evaluates a test set batch after batch, train.py: one call per batch): with 10 calls per repeat the start-up of
    the first call (~100 us of host work before its decoder launch) was 2-3 % of the reported time per call."""
    from mvae_amd import functional as Fn, utils
    from mvae_amd.models import FeedForwardVAE

    class _DS:
        in_dim, img_dims = D, None

        @staticmethod
        def reconstruction_loss(x_, x):
            return Fn.bce_rows(x_, x)

    torch.manual_seed(0)
    m = FeedForwardVAE(H, utils.parse_components(MODEL, False), _DS(), False).to(dev)
    x = (torch.rand(B, D, device=dev) > 0.7).float()
    for _ in range(3):
        out = m.log_likelihood(x, n=n)
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(iters):
            out = m.log_likelihood(x, n=n)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / iters)
    dt = sorted(times)[len(times) // 2]
    assert bool(torch.isfinite(out[0]).all()), "non-finite log-likelihood"
    Z = m.total_z_dim
    flops = 2.0 * n * B * (Z * H + H * D) + 2.0 * B * (D * H + H * 2 * Z)
    tf = flops / dt / 1e12
    # the dominant launch alone (the decoder + per-row BCE of the n * B sampled rows: mvae_decode_bce_rows), HIP events on
    # the stream it is launched on; None if the shape fell back to the three generic operators
    kern = None
    zs = torch.randn(n, B, Z, device=dev)
    if Fn.decode_bce_rows(zs, m.fc_d0.weight, m.fc_d0.bias, m.fc_logits.weight, m.fc_logits.bias, x) is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ks = []
        for _ in range(5):
            e0.record()
            for _ in range(iters):
                Fn.decode_bce_rows(zs, m.fc_d0.weight, m.fc_d0.bias, m.fc_logits.weight, m.fc_logits.bias, x)
            e1.record()
            torch.cuda.synchronize()
            ks.append(e0.elapsed_time(e1) / iters)
        kms = sorted(ks)[len(ks) // 2]
        kfl = 2.0 * n * B * (Z * H + H * D)
        kern = {"kernel": "k_decode_bce_rows: relu(z W_d0^T + b) W_logits^T + b -> BCE row sums, hidden layer and logits on chip",
                "kernel_ms": kms, "kernel_flops": kfl, "kernel_achieved_TFLOPs": kfl / kms / 1e9,
                "kernel_mfma_frac": kfl / kms / 1e9 / F32_MFMA_PEAK_TF,
                "kernel_algorithmic_bytes": 4.0 * (n * B * (Z + 1) + H * (Z + 1) + D * (H + 1) + B * D),
                "counters": "profiles/r05_loglik_decoder_pmc.txt"}
    return {"metric": f"IWAE log-likelihood batches/sec (B={B}, n={n}) MNIST {MODEL}", "value": 1.0 / dt, "unit": "batches/sec",
            "ms_per_batch": dt * 1e3, "n": n, "batch": B, "dtype": "f32", "timed_repeats": 5, "iters_per_repeat": iters,
            "repeat_ms_per_batch": {"median": dt * 1e3, "first": times[0] * 1e3, "min": min(times) * 1e3, "max": max(times) * 1e3},
            "roofline": dict({"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TF,
                              "flops_per_batch": flops, "scope": "the whole log_likelihood call"}, **(kern or {})),
            "baseline_config": "scope row f-1: ModelVAE.log_likelihood (vae.py:82-123), the reference's test-time estimator"}


def configs_summary(line):
    """The last object of the JSON line: every BASELINE config of this run as [value, ms, roofline frac], short enough to
    survive a 2000-character tail of stdout (the full legs are under `configs`)."""
    def row(d, ms_key="ms_per_step"):
        if not isinstance(d, dict) or "error" in d or "value" not in d:
            return None
        fr = (d.get("roofline") or {}).get("frac")
        return [round(d["value"], 1), round(d[ms_key], 5), None if fr is None else round(fr, 4)]
    c = line.get("configs", {})
    out = {"fmt": "[value/s, ms, roofline frac]; median of timed_repeats repeats",
           "h2s2e2": row(line), "e6": row(c.get("e6")), "prod36": row(c.get("prod36")), "conv": row(c.get("conv")),
           "conv_f32_mfma": row(c.get("conv_f32_mfma")), "conv_split": row(c.get("conv_split_bf16_products")),
           "epoch_pipeline": row(c.get("epoch_pipeline")), "epoch_b100": row(c.get("epoch_pipeline_b100")),
           "loglik": row(c.get("loglik"), "ms_per_batch"),
           "conv_mode": (c.get("conv") or {}).get("contraction_mode"),
           "timed_repeats": line["config"]["timed_repeats"],
           "first_repeat": round(line["config"]["first_repeat_value"], 1)}
    return out


def conv_short_leg(mode):
    """A 20-step leg of the conv config in contraction mode `mode` (None: the library's current one), reduced to the fields
    the `configs` block of the headline line carries."""
    from mvae_amd._lib import load as _load
    prev = _load().mvae_set_contraction_mode(-1)
    try:
        if mode is not None:
            _load().mvae_set_contraction_mode(mode)
        leg = conv_leg(20, 5)
    finally:
        _load().mvae_set_contraction_mode(prev)
    out = {k: v for k, v in leg.items() if k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "roofline")}
    out["contraction_mode"] = leg["config"]["contraction_mode"]
    out["timed_repeats"] = leg["config"]["timed_repeats"]
    out["repeat_ms_per_step"] = leg["config"]["repeat_ms_per_step"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--graph-steps", type=int, default=100,
                    help="steps captured per HIP graph for timed regions longer than ONE_GRAPH_MAX steps (shorter "
                         "regions are ONE graph); 0 = eager launches")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed repeats of EXACTLY --steps steps each, every one bracketed by barrier + synchronize "
                         "(value = the median repeat; first / min / max reported beside it); default 5 (SURVEY 8d "
                         "protocol: median of 5 repeats)")
    ap.add_argument("--reset-every", type=int, default=0,
                    help="restore the initial parameters / optimizer state every N steps (0 = never, the default: "
                         "40 000 consecutive learnable-curvature steps on the cycled synthetic batches stay finite); a "
                         "diagnostic for configurations that diverge")
    ap.add_argument("--prewarm", type=int, default=0,
                    help="OPT-IN diagnostic (default 0 = the contract's protocol: exactly --warmup untimed steps, then the "
                         "timed region): replay the captured timed graph this many times, on a snapshot of the model state "
                         "that is restored afterwards, in front of a short (one-graph) timed region.  A device that idled "
                         "since the capture runs its first replays 5-35 %% slow; round 3 did 12 of these by default, which "
                         "made its headline incomparable with the stated warm-up (ADVICE r3) -- now off unless asked for")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short legs of BASELINE configs[0], [3], [4] reported under `configs`")
    ap.add_argument("--model", type=str, default=MODEL,
                    help="latent space string; the driver's metric is the default (BASELINE configs[1]); "
                         "'e6' = configs[0], '6h2,6s2,6e2' = configs[3]")
    ap.add_argument("--fixed-curvature", action="store_true")
    ap.add_argument("--config", type=str, default="mlp", choices=["mlp", "conv"],
                    help="mlp (default): the driver's metric, BASELINE configs[1]; conv: BASELINE configs[4] on one GPU "
                         "(CIFAR shapes, conv architecture, batch 256), same JSON schema with an MFMA roofline")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the GLOBAL batch stays 128 and is split by rows across the ranks (the "
                         "configuration whose summed gradient equals the single-device step, SURVEY section 8d); the "
                         "driver's contract is the default, weak scaling (128 rows per GPU)")
    ap.add_argument("--force-dp", action="store_true",
                    help="diagnostic: take the data-parallel route (gradients -> RCCL all-reduce -> k_optim) even at "
                         "world size 1")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Libraries below (RCCL's version banner, HIP runtime notices) write to
    # the C-level stdout, buffered until exit: keep the real stdout aside for the JSON line and point fd 1 at stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        os.close(json_fd)

    if args.config == "conv":
        if args.steps == 2000 and args.warmup == 200:  # the MLP defaults: a conv step is ~40x longer
            args.steps, args.warmup = 200, 20
        world, rank, local_rank, dist_on, backend = init_dist(args.force_dp)
        line = conv_leg(args.steps, args.warmup, graph=args.graph_steps > 0, world=world, rank=rank, dist_on=dist_on,
                        backend=backend, strong=args.strong, force_dp=args.force_dp, repeats=args.repeats or 5)
        if rank == 0:
            emit(line)
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    world, rank, local_rank, dist_on, backend = init_dist(args.force_dp)
    if dist_on:
        import torch.distributed as dist
    dev = torch.device("cuda", local_rank)

    from mvae_amd import synthetic
    from mvae_amd.engine import StepEngine
    from mvae_amd.runner import StepRunner

    comps = parse_model(args.model)
    eng = StepEngine(comps, D, H, dev, radius_trainable=[not args.fixed_curvature] * len(comps), lr=1e-3)
    shapes = [(n, s) for n, _, s in eng.flat.entries]
    eng.load_state(synthetic.synthetic_state(shapes, radius=2.0))
    # Every timed step is part of a graph replay; a timed region of up to ONE_GRAPH_MAX steps (the driver's
    # `--steps 20 --warmup 5`) is ONE graph, the warm-up another.
    gs, plan = make_plan(args.steps, args.warmup, args.graph_steps)
    n_data = min(256, args.steps + args.warmup) if plan is not None else max(gs, 1) * 2  # resident batches, cycled
    if args.strong and world > 1:
        from mvae_amd.distributed import shard_rows
        lo, hi = shard_rows(B, rank, world)  # every rank builds the same global batches and keeps its own rows
        xs = synthetic.digits_like_batches(n_data, B, seed=4321)[:, lo:hi].contiguous().to(dev)
        eps = synthetic.eps_batches(n_data, B, eng.layout.eps_dim, rank=0)[:, lo:hi].contiguous().to(dev)
    else:
        xs = synthetic.digits_like_batches(n_data, B, seed=4321 + rank).to(dev)
        eps = synthetic.eps_batches(n_data, B, eng.layout.eps_dim, rank=rank).to(dev)
    strong = bool(args.strong and world > 1)
    runner = StepRunner(eng, xs, eps, beta=1.0, do_curvature_step=not args.fixed_curvature,
                        graph_steps=gs, graph_plan=plan,
                        world_size=world, reset_every=args.reset_every, force_exchange=args.force_dp)

    if (runner.capture_failed or os.environ.get("MVAE_BENCH_FAKE_CAPTURE_FAILURE")) and world > 1 and \
            not os.environ.get("MVAE_BENCH_REEXEC"):
        # An invalidated capture can leave HIP unusable for the rest of the process (every later call reports the
        # capture error).  Rather than lose the measurement, every rank starts over as a fresh process image without
        # graphs (all ranks fail alike, so they re-rendezvous; a new port, the old store's socket may still be open).
        print("[bench] capture of the exchange failed: restarting without HIP graphs", file=sys.stderr, flush=True)
        os.environ["MVAE_BENCH_REEXEC"] = "1"
        os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29531")) + 17)
        os.environ["TORCHELASTIC_USE_AGENT_STORE"] = "False"  # rank 0 hosts a fresh store (the agent's holds stale keys)
        os.dup2(json_fd, 1)
        os.execv(sys.executable, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--graph-steps", "0"])

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    repeats = args.repeats if args.repeats > 0 else 5
    prewarm = 0
    if plan is not None and runner.gs > 0 and args.prewarm > 0:
        # Opt-in only (--prewarm N).  A short timed region (the driver's 20 steps = 0.65 ms) starts on a device that has
        # been idle since the capture: the first replays after an idle period run 5-35 % slower than the steady state
        # (clock / power ramp; tools/probe_short_run.py: 44, 33.9, 33.2, ... 32.5 us/step over the first nine 20-step
        # replays).  With --prewarm the captured timed graph is replayed on a snapshot of the model state, which is
        # restored before the contractual warm-up; the line then says so (config.prewarm_replays > 0).
        keep = [t.clone() for t in (eng.params, eng.adam_m, eng.adam_v, eng.counters, eng.stats)]
        timed_graph = runner.graphs[args.warmup if args.warmup > 0 else 0][0]
        while prewarm < args.prewarm:
            timed_graph.replay()
            torch.cuda.synchronize()
            prewarm += 1
        for dst, src in zip((eng.params, eng.adam_m, eng.adam_v, eng.counters, eng.stats), keep):
            dst.copy_(src)
        sync_all()
    runner.run(args.warmup)
    sync_all()
    times = []
    replays_before, gsteps_before = runner.replays, runner.graph_steps_replayed
    for _ in range(repeats):
        if plan is not None and runner.gs > 0 and args.warmup > 0:
            runner.cursor = args.warmup  # a one-graph timed region: every repeat replays THE timed graph (same batches)
        t0 = time.perf_counter()
        runner.run(args.steps)  # EXACTLY --steps steps per timed region
        sync_all()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        times.append(dt)
    dt = sorted(times)[len(times) // 2]  # the median repeat
    graph_replays = (runner.replays - replays_before) // repeats
    graph_steps_replayed = (runner.graph_steps_replayed - gsteps_before) // repeats
    stats = eng.read_stats()
    assert stats["sum"]["steps"] == args.warmup + repeats * args.steps + runner.capture_steps, stats["sum"]["steps"]
    finite = stats["last"]["elbo"] == stats["last"]["elbo"] and abs(stats["last"]["elbo"]) != float("inf")
    assert finite or os.environ.get("MVAE_BENCH_ALLOW_NONFINITE"), "non-finite ELBO"

    ranks_identical, peer_timeouts = None, None
    if dist_on:
        # the replicas must hold the same parameters after the run (the exchange is the only thing keeping them equal),
        # and a peer route must not have given up a single wait: host-side checks, after the timed region
        import hashlib
        digest = hashlib.sha1(eng.params.cpu().numpy().tobytes()).hexdigest()
        digests = [None] * world
        dist.all_gather_object(digests, digest)
        ranks_identical = all(d_ == digests[0] for d_ in digests)
        assert ranks_identical, f"the ranks' parameters differ after the run: {digests}"
        if runner.dp is not None and runner.dp.peer is not None:
            tmo = [None] * world
            dist.all_gather_object(tmo, runner.dp.peer.timeouts())
            peer_timeouts = int(sum(tmo))
            assert peer_timeouts == 0, f"peer exchange: waits timed out per rank {tmo}"
    if rank != 0:
        if dist_on:  # leave together with rank 0 (which still profiles the launches): no rank tears the group down early
            dist.barrier()
            dist.destroy_process_group()
        return

    # per-launch durations measured live with HIP events on the launch stream (mvae_step_profile)
    prof = eng.profile_step(xs[0], eps[0], 1.0, not args.fixed_curvature, iters=200)
    roof = mlp_roofline(eng, prof, dt / args.steps, args.fixed_curvature,
                        pmc_config={"e6": "e6", "6h2,6s2,6e2": "prod36", "h40": "h40"}.get(args.model))

    line = {
        "metric": f"ELBO-steps/sec (batch 128) MNIST {args.model}",
        "value": args.steps * (1 if strong else world) / dt,
        "unit": "ELBO-steps/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: " if args.model == MODEL and not args.fixed_curvature else "") +
                               f"MNIST shapes (D=784), model {args.model}, "
                               f"{'fixed' if args.fixed_curvature else 'learnable'} curvature, "
                               "MLP h_dim=400, " + ("global batch 128 split by rows" if strong else "batch 128 per GPU") +
                               ", epoch>=10 state",
                   "global_batch": B if strong else B * world, "parallelism": f"dp{world}" + ("(forced exchange)" if args.force_dp else ""),
                   "exchange": ((runner.dp.exchange + (" + sharded optimizer" if getattr(runner.dp, "shard", False) else "") +
                                 (f" ({runner.dp.exchange_note})" if runner.dp.exchange_note else ""))
                                if runner.dp is not None else "none (fused optimizer epilogues)"),
                   "ranks_identical": ranks_identical, "peer_timeouts": peer_timeouts,
                   "graph_steps": runner.gs,
                   "graph_replays": graph_replays,
                   "steps_in_graph_replays": graph_steps_replayed,
                   "timed_repeats": repeats,
                   "prewarm_replays": prewarm,  # of the timed graph, on a snapshot of the state that is restored
                   "repeat_ms_per_step": {"median": dt / args.steps * 1e3, "first": times[0] / args.steps * 1e3,
                                          "min": min(times) / args.steps * 1e3, "max": max(times) / args.steps * 1e3,
                                          "all": [round(t / args.steps * 1e3, 6) for t in times]},
                   "first_repeat_value": args.steps * (1 if strong else world) / times[0],
                   "inputs": "x and eps resident in HBM before the timed region (SURVEY 8d); the device-side gather + "
                             "binarisation + eps draw of mvae_prepare_batch is NOT in the timed step",
                   "state_reset_every": args.reset_every,
                   "final_elbo_per_sample": stats["last"]["elbo"] / xs.shape[1]},
        "roofline": roof,
    }
    default_workload = args.model == MODEL and not args.fixed_curvature and not args.force_dp
    if world == 1 and default_workload and not args.no_extra_configs:
        # the other single-GPU BASELINE configs, short legs on the same box in the same run (configs[0..4])
        del runner, eng
        extra = {}
        for key, fn in (("e6", lambda: mlp_leg("e6", True, 400, 40, dev)),
                        ("prod36", lambda: mlp_leg("6h2,6s2,6e2", False, 400, 40, dev)),
                        ("conv", lambda: conv_short_leg(None))):
            try:
                extra[key] = fn()
            except Exception as e:  # noqa: BLE001  (a failing leg must not lose the headline line)
                extra[key] = {"error": f"{type(e).__name__}: {e}"}
        # configs.conv is the library's DEFAULT contraction mode (2: exact-f32 forward, split-bf16 backward; per-entry 1e-4
        # against the reference's own step, tests/test_conv_gpu.py::test_conv_step_b256_vs_the_reference).  Beside it, the
        # same step with the f32-input MFMA everywhere (mode 0) and with split products everywhere (mode 1, whose
        # step-level gradient bar is 1e-3: its forward rounding can flip a ReLU output).
        for key, m in (("conv_f32_mfma", 0), ("conv_split_bf16_products", 1)):
            try:
                extra[key] = conv_short_leg(m)
            except Exception as e:  # noqa: BLE001
                extra[key] = {"error": f"{type(e).__name__}: {e}"}
        try:
            extra["epoch_pipeline"] = epoch_pipeline_leg(dev)
        except Exception as e:  # noqa: BLE001
            extra["epoch_pipeline"] = {"error": f"{type(e).__name__}: {e}"}
        try:  # the reference CLI's default batch size: padding rows keep it on the fused kernels
            extra["epoch_pipeline_b100"] = epoch_pipeline_leg(dev, batch=100)
        except Exception as e:  # noqa: BLE001
            extra["epoch_pipeline_b100"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            extra["loglik"] = loglik_leg(dev)
        except Exception as e:  # noqa: BLE001
            extra["loglik"] = {"error": f"{type(e).__name__}: {e}"}
        extra["e6"].setdefault("baseline_config", "configs[0]: MNIST e6, fixed curvature")
        extra["prod36"].setdefault("baseline_config", "configs[3]: 6h2,6s2,6e2 (36-dim latent), learnable curvature")
        extra["conv"].setdefault("baseline_config", "configs[4] on ONE GPU: CIFAR conv h_dim=8192, batch 256")
        line["configs"] = extra
    if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N = 1 only
        line["cpu_baseline"] = cpu_baseline(model=args.model, fixed=args.fixed_curvature)
    line["configs_summary"] = configs_summary(line)  # LAST: the tail of stdout carries every config's number
    emit(line)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
