"""CPU-only: the oracle's trajectory over many steps on the bench's resident batches (diagnostic)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvae_amd import synthetic  # noqa: E402
from oracle import model as M  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
torch.set_num_threads(8)
spec = M.Spec("h2,s2,e2", in_dim=784, h_dim=400, fixed_curvature=False)
orc = M.StepOracle(spec, synthetic.synthetic_state(spec.named_shapes(), radius=2.0))
xs = synthetic.digits_like_batches(200, 128) if os.environ.get('DIGITS') else synthetic.binary_batches(200, 128, 784)
eps = synthetic.eps_batches(200, 128, 6)
t0 = time.time()
for s in range(steps):
    out = orc.train_step(xs[s % 200], eps[s % 200], 1.0, epoch=12)
    if (s + 1) % every == 0 or not torch.isfinite(out.elbo):
        print(f"step {s+1} elbo/sample {float(out.elbo)/128:.4f} kl {[round(float(k),4) for k in out.kl.sum(1)]} "
              f"R_h {float(orc.P['components.0._nradius']):.6f} R_s {float(orc.P['components.1._pradius']):.6f} "
              f"t={time.time()-t0:.0f}s", flush=True)
        if not torch.isfinite(out.elbo):
            break
