"""Dev: per-tensor gradient error of the conv step against the float64 oracle for several batch sizes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvae_amd import synthetic
from mvae_amd.conv import ConvEngine
from oracle import model as M
dev = torch.device("cuda:0")
spec = M.Spec("h2,s2,e2", in_dim=3072, h_dim=8192, arch="conv", fixed_curvature=False)
state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0, transposed_conv=("d1", "d2", "d3"))
for B in [int(a) for a in sys.argv[1:]] or [32, 256]:
    x = synthetic.uniform_batches(1, B, 3072)[0]
    eps = synthetic.eps_batches(1, B, 6)[0]
    orc = M.StepOracle(spec, state0, dtype=torch.float64)
    ref = orc.train_step(x.double(), eps.double(), beta=1.0, epoch=12)
    eng = ConvEngine([("h", 2), ("s", 2), ("e", 2)], dev, radius_trainable=[True] * 3)
    eng.load_state(state0)
    out = eng.forward_backward(x.to(dev), eps.to(dev), 1.0, want_outputs=True)
    print(f"B={B}")
    for n, t in eng.grad_views().items():
        if orc.P[n].grad is None:
            continue
        a, b = t.detach().cpu().double().numpy(), orc.P[n].grad.numpy()
        print(f"  {n:32s} max|b| {np.abs(b).max():.3e}  max|a-b| {np.abs(a-b).max():.3e}  rel {np.abs(a-b).max()/np.abs(b).max():.2e}")
