import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import synthetic
from mvae_amd.engine import StepEngine
from oracle import model as M
dev = torch.device("cuda:0")
model, B = "u10,p40,d40", 16
spec = M.Spec(model, in_dim=784, h_dim=400, fixed_curvature=False)
state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0)
x = synthetic.binary_batches(1, B, 784)[0]; eps = synthetic.eps_batches(1, B, spec.total_true_dim)[0]
ref = M.StepOracle(spec, state0).train_step(x, eps, beta=0.7, epoch=12)
ref64 = M.StepOracle(spec, state0, dtype=torch.float64).train_step(x.double(), eps.double(), beta=0.7, epoch=12)
comps = [(c.letter, c.true_dim) for c in spec.components]
outs = {}
for nc in ("0", "1"):
    os.environ["MVAE_NO_COOP"] = nc
    eng = StepEngine(comps, 784, 400, dev, radius_trainable=[True] * len(comps))
    eng.load_state(state0)
    out = eng.forward_backward(x.to(dev), eps.to(dev), 0.7, want_outputs=True)
    outs[nc] = out["kl"].cpu().numpy()
r = ref.kl.detach().numpy()
print("oracle f32 ", r[0, 6:10])
try:
    print("oracle f64 ", ref64.kl.detach().numpy()[0, 6:10])
except Exception as e:
    print("f64 oracle failed", e)
print("coop       ", outs["0"][0, 6:10])
print("per-lane   ", outs["1"][0, 6:10])
