"""BASELINE config [4] (CIFAR shapes, conv architecture, h_dim 8192, batch 256): steps/s of the HIP path on one GPU.
Not the driver's bench (bench.py is config [1]); prints one JSON line.  The CPU oracle's time for the same step is
measured by tests/dev/conv_cpu_time.py (only tests/ may import the oracle)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import synthetic
from mvae_amd.conv import ConvEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
eng = ConvEngine([("h", 2), ("s", 2), ("e", 2)], dev, radius_trainable=[True, True, False])
shapes = [(name, tuple(v.shape)) for name, v in eng.param_views().items()]
eng.load_state(synthetic.synthetic_state(shapes, radius=2.0, transposed_conv=("d1", "d2", "d3")))
xs = synthetic.uniform_batches(4, B, 3072).to(dev)
eps = synthetic.eps_batches(4, B, 6).to(dev)
for i in range(5):
    eng.train_step(xs[i % 4], eps[i % 4], 1.0, True)
torch.cuda.synchronize()
graph = None
if os.environ.get("GRAPH") == "1":  # the timed region as ONE HIP graph (what bench.py times), replayed twice: warm-up + timed
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(steps):
            eng.train_step(xs[i % 4], eps[i % 4], 1.0, True)
    graph.replay()
    torch.cuda.synchronize()
t0 = time.perf_counter()
if graph is not None:
    graph.replay()
else:
    for i in range(steps):
        eng.train_step(xs[i % 4], eps[i % 4], 1.0, True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
elbo = eng.read_stats()["last"]["elbo"] / B
flops = 79.5e9 * B / 256
print(json.dumps({"workload": f"conv h2,s2,e2 h_dim=8192 B={B}", "steps_per_s": steps / dt, "ms_per_step": dt / steps * 1e3,
                  "tflops": flops * steps / dt / 1e12, "elbo_per_sample": elbo}))
