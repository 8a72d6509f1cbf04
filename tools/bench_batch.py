"""Dev tool: ELBO-steps/s and samples/s of the fused step for several batch sizes (h2,s2,e2, MNIST shapes)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import synthetic
from mvae_amd.engine import StepEngine
from mvae_amd.runner import StepRunner
dev = torch.device("cuda:0")
for B in [int(a) for a in sys.argv[1:]] or [100, 128, 256, 512, 1024]:
    eng = StepEngine([("h", 2), ("s", 2), ("e", 2)], 784, 400, dev, radius_trainable=[True, True, False])
    eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
    xs = synthetic.digits_like_batches(100, B).to(dev)
    eps = synthetic.eps_batches(100, B, 6).to(dev)
    r = StepRunner(eng, xs, eps, beta=1.0, do_curvature_step=True, graph_steps=50, reset_every=2000)
    r.run(500)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.run(4000)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"B={B:5d}  {4000 / dt:9.0f} steps/s  {4000 * B / dt / 1e6:7.2f} M samples/s  {dt / 4000 * 1e6:7.1f} us/step  "
          f"elbo/sample {eng.read_stats()['last']['elbo'] / B:9.2f}")
