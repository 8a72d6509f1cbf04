"""bench.py prints exactly ONE JSON line on stdout with the fields the driver reads (single-GPU and forced exchange)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))  # a free rendezvous port per run
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "50", *flags]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]  # no retry: the exchange route has no watchdog thread to lose a capture to
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # library banners must not reach stdout
    return json.loads(lines[0])


def test_bench_line_schema():
    d = _run("--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 300 and d["warmup"] == 50 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 300 / (d["ms_per_step"] * 1e-3 * 300)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["kernel"] in r["kernel_ms"] and set(r["per_kernel"]) == set(r["kernel_ms"])
    assert d["value"] > 1000  # steps/s: a silent eager fallback would be two orders of magnitude below the HIP path
    assert d["config"]["graph_replays"] == 1 and d["config"]["steps_in_graph_replays"] == 300  # ONE replay is timed
    # the other single-GPU BASELINE configs ride in the same line (configs[0..4])
    for key in ("e6", "prod36", "conv"):
        leg = d["configs"][key]
        assert "error" not in leg, leg
        assert leg["value"] > 100 and leg["ms_per_step"] > 0 and 0 < leg["roofline"]["frac"] < 1, leg
    # whole epochs through the device-side input pipeline (scope row f-2): every step prepares its successor's inputs
    ep = d["configs"]["epoch_pipeline"]
    assert "error" not in ep, ep
    assert ep["value"] > 1000 and ep["steps"] == ep["epochs"] * ep["steps_per_epoch"] and "in_step_preparation=True" in ep["workload"]
    # median of 5 timed repeats of exactly --steps steps; the first repeat is reported beside it
    assert d["config"]["timed_repeats"] == 5 and len(d["config"]["repeat_ms_per_step"]["all"]) == 5
    rp = d["config"]["repeat_ms_per_step"]
    assert rp["min"] <= rp["median"] <= rp["max"] and abs(rp["median"] - d["ms_per_step"]) < 1e-9
    # scope row f-1 as a leg of its own, and the compact summary as the LAST key of the line (what a truncated tail keeps)
    ll = d["configs"]["loglik"]
    assert "error" not in ll and ll["ms_per_batch"] > 0 and 0 < ll["roofline"]["frac"] < 1, ll
    # its dominant launch (the fused decoder + BCE) timed alone: below the whole call, above half of the f32 MFMA peak
    assert 0 < ll["roofline"]["kernel_ms"] < ll["ms_per_batch"] and 0.5 < ll["roofline"]["kernel_mfma_frac"] < 1, ll["roofline"]
    assert list(d)[-1] == "configs_summary"
    cs = d["configs_summary"]
    assert len(json.dumps(cs)) <= 700, len(json.dumps(cs))
    for key in ("h2s2e2", "e6", "prod36", "conv", "conv_f32_mfma", "conv_split", "epoch_pipeline", "epoch_b100", "loglik"):
        assert cs[key] is not None and cs[key][0] > 0, (key, cs)
    # the reference CLI's default batch size rides on the fused kernels through padding rows (112-row buffers)
    b100 = d["configs"]["epoch_pipeline_b100"]
    assert b100["batch"] == 100 and b100["buffer_rows"] == 112 and b100["kernel_path"] == "fused", b100
    assert abs(cs["h2s2e2"][0] - d["value"]) < 0.06 and cs["timed_repeats"] == 5


def test_bench_forced_exchange_route():
    """The data-parallel route (gradients -> ncclAllReduce on librccl directly, on the step's streams -> k_optim) captured
    into ONE HIP graph, at world size 1: no ProcessGroupNCCL, so the capture cannot be lost to a watchdog thread."""
    d = _run("--no-cpu-baseline", "--force-dp")
    assert "forced exchange" in d["config"]["parallelism"] and d["config"]["exchange"] == "rccl"
    assert d["config"]["graph_steps"] == 300 and d["config"]["graph_replays"] == 1
    assert d["value"] > 1000
