"""The CPU oracle's time for one step of BASELINE config [4] (conv architecture, h_dim 8192, batch 256) on the host
cores -- the number quoted next to tools/bench_conv.py's GPU figure."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvae_amd import synthetic
from oracle import model as M

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
spec = M.Spec("h2,s2,e2", in_dim=3072, h_dim=8192, arch="conv", fixed_curvature=False)
state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0, transposed_conv=("d1", "d2", "d3"))
torch.set_num_threads(min(16, os.cpu_count() or 1))
orc = M.StepOracle(spec, state0)
xs, eps = synthetic.uniform_batches(4, B, 3072), synthetic.eps_batches(4, B, 6)
orc.train_step(xs[0], eps[0], 1.0, epoch=12)
t0 = time.perf_counter()
for i in range(n):
    orc.train_step(xs[i % 4], eps[i % 4], 1.0, epoch=12)
dt = (time.perf_counter() - t0) / n
print(json.dumps({"workload": f"conv h2,s2,e2 h_dim=8192 B={B} (CPU oracle)", "ms_per_step": dt * 1e3,
                  "threads": torch.get_num_threads()}))
