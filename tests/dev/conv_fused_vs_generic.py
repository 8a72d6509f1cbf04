"""Dev probe: ConvEngine step with MVAE_CONV_FUSED=1 against =0, every output / gradient."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvae_amd import synthetic
from mvae_amd.conv import ConvEngine
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
comps = [("h", 2), ("s", 2), ("e", 2)]
x = synthetic.uniform_batches(1, B, 3072)[0].to(dev)

def run(fused):
    os.environ["MVAE_CONV_FUSED"] = fused
    eng = ConvEngine(comps, dev, radius_trainable=[True] * 3)
    shapes = [(name, tuple(v.shape)) for name, v in eng.param_views().items()]
    eng.load_state(synthetic.synthetic_state(shapes, radius=1.7, transposed_conv=("d1", "d2", "d3")))
    eps = synthetic.eps_batches(1, B, 6)[0].to(dev)
    c = eng._forward(x, eps)
    out = eng.forward_backward(x, eps, 0.7, want_outputs=True)
    torch.cuda.synchronize()
    return eng, out, c

ef, of, cf = run("1")
eg, og, cg = run("0")
rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for k in ("heads", "z", "kl", "t0", "b1", "logits"):
    print(k, rel(cf[k], cg[k]), end=" | ")
print("mask flips", int(((cf["t0"] > 0) != (cg["t0"] > 0)).sum()), "zeros", int((cg["t0"] == 0).sum()), "of", cg["t0"].numel())
for k in ("a0", "a1", "a2", "t0", "b1", "b2"):
    print("activation sign flips", k, int(((cf[k] > 0) != (cg[k] > 0)).sum()), "of", cf[k].numel(),
          "largest flipped value", float(torch.where((cf[k] > 0) != (cg[k] > 0), (cf[k] - cg[k]).abs(), torch.zeros_like(cf[k])).max()))
for (n, a), (_, b) in zip(ef.grad_views().items(), eg.grad_views().items()):
    print(f"{n:40s} {rel(a, b):.3e}")
