"""Dev tool: HIP step and CPU oracle in lock-step through the radius warm-up (train.py:189-194) on the same x / eps:
where does either go non-finite, and how far apart are the ELBOs before that?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvae_amd import synthetic
from mvae_amd.engine import StepEngine
from oracle import model as M

model = sys.argv[1] if len(sys.argv) > 1 else "h2,s2,e2"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps_per_epoch = int(sys.argv[3]) if len(sys.argv) > 3 else 469
with_oracle = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
dev = torch.device("cuda:0")
torch.manual_seed(int(sys.argv[5]) if len(sys.argv) > 5 else 0)
spec = M.Spec(model, in_dim=784, h_dim=400, fixed_curvature=False)
state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0)
comps = [(c.letter, c.true_dim) for c in spec.components]
eng = StepEngine(comps, 784, 400, dev, radius_trainable=[l != "e" for l, _ in comps], lr=1e-3)
eng.load_state(state0)
orc = M.StepOracle(spec, state0) if with_oracle else None
torch.set_num_threads(8)
n_data = 64
xs = synthetic.digits_like_batches(n_data, 128, seed=77)
eps = synthetic.eps_batches(n_data, 128, eng.layout.eps_dim)
xg, eg = xs.to(dev), eps.to(dev)
step = 0
for epoch in range(epochs):
    if epoch < 10:
        eng.set_radii(float(11 - epoch))
        if orc is not None:
            orc.begin_epoch(epoch)
    for s in range(steps_per_epoch):
        i = step % n_data
        if not with_oracle:  # fresh noise every step (the oracle leg keeps the fixed eps set for the lock-step)
            eg[i].normal_()
        eng.train_step(xg[i], eg[i], 1.0, epoch >= 10)
        e_gpu = eng.read_stats()["last"]["elbo"] / 128
        e_cpu = float("nan")
        if orc is not None:
            out = orc.train_step(xs[i], eps[i], 1.0, epoch=epoch)
            e_cpu = float(out.elbo) / 128 if hasattr(out, "elbo") else float("nan")
        if step % 100 == 0 or not np.isfinite(e_gpu) or (orc is not None and not np.isfinite(e_cpu)):
            print(f"epoch {epoch} step {step}: elbo/sample gpu {e_gpu:.4f} cpu {e_cpu:.4f}", flush=True)
        if not np.isfinite(e_gpu) or (orc is not None and not np.isfinite(e_cpu)):
            st = eng.read_stats()["last"]
            if not with_oracle:  # replay the failing batch forward-only from the state BEFORE the step is impossible
                pass             # (in-place update); the per-component sums below say which component broke
            print("gpu component_kl", st["component_kl"], "bce", st["bce"])
            for name, v in eng.param_views().items():
                if not bool(torch.isfinite(v).all()):
                    print("  non-finite parameter:", name)
            sys.exit(0)
        step += 1
print("finite throughout")
