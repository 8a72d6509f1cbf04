"""Dev probe: run the two-call path (gradients only + k_optim) so that rocprof shows the kernels without Adam epilogues."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import synthetic
from mvae_amd.engine import StepEngine
dev = torch.device("cuda:0")
eng = StepEngine([("h", 2), ("s", 2), ("e", 2)], 784, 400, dev, radius_trainable=[True, True, False])
eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
xs = synthetic.digits_like_batches(8, 128).to(dev); eps = synthetic.eps_batches(8, 128, 6).to(dev)
for i in range(600):
    eng.forward_backward(xs[i % 8], eps[i % 8], 1.0)
    eng.optimizer_step(True, batch=128)
torch.cuda.synchronize()
print("done", eng.read_stats()["last"]["elbo"] / 128)
