"""Dev tool: what a SHORT timed region (the driver's --steps 20 --warmup 5) costs on top of the steady-state step time.
Replays the 20-step graph back to back and one at a time, prints host-side and event-side durations per replay."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import synthetic
from mvae_amd.engine import StepEngine
from mvae_amd.runner import StepRunner
import bench
dev = torch.device("cuda:0")
comps = bench.parse_model("h2,s2,e2")
eng = StepEngine(comps, 784, 400, dev, radius_trainable=[True] * 3, lr=1e-3)
eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
K, W = 20, 5
xs = synthetic.digits_like_batches(K + W, 128, seed=4321).to(dev)
eps = synthetic.eps_batches(K + W, 128, eng.layout.eps_dim, rank=0).to(dev)
runner = StepRunner(eng, xs, eps, beta=1.0, do_curvature_step=True, graph_steps=K, graph_plan=[W, K])
g5, g20 = runner.graphs[0][0], runner.graphs[W][0]
torch.cuda.synchronize()
out = []
time.sleep(0.5)  # idle GPU, like a fresh process right after capture
for rep in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    g20.replay()
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.append((dt * 1e6 / K, e0.elapsed_time(e1) * 1e3 / K))
print("single replays after idle: (host us/step, event us/step)")
print(" ".join(f"{a:.1f}/{b:.1f}" for a, b in out))
# back-to-back: N replays then one sync
for n in (1, 2, 5, 20, 100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g20.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n:4d} replays back to back: {dt * 1e6 / (n * K):.2f} us/step")
# idle gaps: does the clock drop after a short idle?
for idle in (0.0, 0.001, 0.01, 0.1, 1.0):
    ts = []
    for rep in range(5):
        for _ in range(50):
            g20.replay()
        torch.cuda.synchronize()
        time.sleep(idle)
        t0 = time.perf_counter()
        g20.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6 / K)
    print(f"after {idle * 1e3:6.1f} ms idle: single 20-step replay {min(ts):.2f} .. {max(ts):.2f} us/step")
# the cost of an empty-ish sync
t0 = time.perf_counter()
for _ in range(100):
    torch.cuda.synchronize()
print(f"synchronize on an idle device: {(time.perf_counter() - t0) * 1e4:.2f} us")
