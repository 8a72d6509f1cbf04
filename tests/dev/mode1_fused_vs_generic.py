"""Dev probe: fused vs generic conv step (MVAE_CONV_FUSED=1 / 0) in the contraction mode of MVAE_CONV_SPLIT_BF16: per-tensor
differences of the forward caches and of every gradient."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvae_amd import synthetic
from mvae_amd.conv import ConvEngine
dev = torch.device("cuda:0")
B = 256
comps = [("h", 2), ("s", 2), ("e", 2)]
x = synthetic.uniform_batches(1, B, 3072)[0].to(dev)
eps = synthetic.eps_batches(1, B, 6)[0].to(dev)
def run(fused):
    os.environ["MVAE_CONV_FUSED"] = fused
    eng = ConvEngine(comps, dev, radius_trainable=[True] * 3)
    shapes = [(name, tuple(v.shape)) for name, v in eng.param_views().items()]
    eng.load_state(synthetic.synthetic_state(shapes, radius=1.7, transposed_conv=("d1", "d2", "d3")))
    acts = eng._forward(x, eps)
    out = eng.forward_backward(x, eps, 0.7, want_outputs=True)
    torch.cuda.synchronize()
    return eng, out, acts
ef, of, cf = run("1")
eg, og, cg = run("0")
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
for k in ("a0", "a1", "a2", "heads", "z", "kl", "t0", "b1", "b2", "logits"):
    if k in cf and k in cg and cf[k] is not None and cg[k] is not None and cf[k].shape == cg[k].shape:
        print(f"act {k:8s} max-rel {rel(cf[k], cg[k]):.3e} flips {int(((cf[k] > 0) != (cg[k] > 0)).sum())}")
for (n, a), (_, b) in zip(ef.grad_views().items(), eg.grad_views().items()):
    print(f"grad {n:34s} max-rel {rel(a, b):.3e}")
for n in ("d2.weight", "d0.weight", "d1.weight", "components.2.fc_mean.weight"):
    a, b = ef.grad_views()[n].double().flatten(), eg.grad_views()[n].double().flatten()
    d = (a - b).abs() / b.abs().max()
    idx = torch.argsort(d, descending=True)[:6]
    print(n, "entries > 1e-4:", int((d > 1e-4).sum()), "of", d.numel(), "rel-l2", float((a - b).norm() / b.norm()),
          [(int(i), float(a[i]), float(b[i])) for i in idx[:4]])
# which of the two is closer to the CPU restatement (tests/dev only: the oracle is test infrastructure)
from oracle import model as M
spec = M.Spec("h2,s2,e2", in_dim=3072, h_dim=8192, arch="conv", fixed_curvature=False)
state0 = synthetic.synthetic_state(spec.named_shapes(), radius=1.7, transposed_conv=("d1", "d2", "d3"))
orc = M.StepOracle(spec, state0)
ref = orc.train_step(x.cpu(), eps.cpu(), beta=0.7, epoch=12)
for n in ("d3.weight", "d2.weight", "d2.bias", "d1.weight", "d0.weight", "components.2.fc_mean.weight", "e2.weight"):
    r = orc.P[n].grad.double().flatten()
    for tag, e in (("fused", ef), ("generic", eg)):
        a = e.grad_views()[n].double().flatten().cpu()
        d = (a - r).abs() / r.abs().max()
        print(f"{n:30s} {tag:8s} vs oracle: entries > 1e-4: {int((d > 1e-4).sum()):6d} of {d.numel()}  max {float(d.max()):.2e}  rel-l2 {float((a - r).norm() / r.norm()):.2e}")
