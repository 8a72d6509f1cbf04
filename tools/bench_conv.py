"""BASELINE config [4] (CIFAR shapes, conv architecture, h_dim 8192, batch 256): steps/s of the HIP path on one GPU
and of the oracle on the host.  Not the driver's bench (bench.py is config [1]); prints one JSON line."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import synthetic
from mvae_amd.conv import ConvEngine
from oracle import model as M

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ncpu = int(sys.argv[3]) if len(sys.argv) > 3 else 3  # oracle steps timed on the host (0: skip)
dev = torch.device("cuda:0")
spec = M.Spec("h2,s2,e2", in_dim=3072, h_dim=8192, arch="conv", fixed_curvature=False)
state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0, transposed_conv=("d1", "d2", "d3"))
eng = ConvEngine([("h", 2), ("s", 2), ("e", 2)], dev, radius_trainable=[True, True, False])
eng.load_state(state0)
xs = synthetic.uniform_batches(4, B, 3072).to(dev)
eps = synthetic.eps_batches(4, B, 6).to(dev)
for i in range(5):
    eng.train_step(xs[i % 4], eps[i % 4], 1.0, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    eng.train_step(xs[i % 4], eps[i % 4], 1.0, True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
elbo = eng.read_stats()["last"]["elbo"] / B
torch.set_num_threads(min(16, os.cpu_count() or 1))
orc = M.StepOracle(spec, state0)
xc, ec = xs.cpu(), eps.cpu()
dtc = float("nan")
if ncpu > 0:
    orc.train_step(xc[0], ec[0], 1.0, epoch=12)
    t1 = time.perf_counter()
    for i in range(ncpu):
        orc.train_step(xc[i % 4], ec[i % 4], 1.0, epoch=12)
    dtc = (time.perf_counter() - t1) / ncpu
flops = 79.5e9 * B / 256
print(json.dumps({"workload": f"conv h2,s2,e2 h_dim=8192 B={B}", "steps_per_s": steps / dt, "ms_per_step": dt / steps * 1e3,
                  "tflops": flops * steps / dt / 1e12, "elbo_per_sample": elbo,
                  "cpu_oracle_ms_per_step": (dtc * 1e3 if ncpu > 0 else None), "cpu_threads": torch.get_num_threads()}))
