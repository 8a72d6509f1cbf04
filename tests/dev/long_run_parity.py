"""Long-run trajectory check: N steps of the fused HIP step vs the oracle on the same resident batches.
Prints ELBO/sample and the radii every `--every` steps.  (Diagnostic; the pass/fail parity tests live in tests/.)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from mvae_amd import synthetic  # noqa: E402
from mvae_amd.engine import StepEngine  # noqa: E402
from oracle import model as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--every", type=int, default=200)
ap.add_argument("--curv-after", type=int, default=0, help="curvature SGD starts after this many steps")
ap.add_argument("--n-data", type=int, default=200)
ap.add_argument("--no-oracle", action="store_true")
ap.add_argument("--model", type=str, default="h2,s2,e2")
ap.add_argument("--digits", action="store_true")
args = ap.parse_args()

torch.set_num_threads(8)
dev = torch.device("cuda:0")
spec = M.Spec(args.model, in_dim=784, h_dim=400, fixed_curvature=False)
state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0)
xs = synthetic.digits_like_batches(args.n_data, 128) if args.digits else synthetic.binary_batches(args.n_data, 128, 784)
eps = synthetic.eps_batches(args.n_data, 128, spec.total_true_dim)
comps = [(c.letter, c.true_dim) for c in spec.components]
eng = StepEngine(comps, 784, 400, dev, radius_trainable=[True] * len(comps))
eng.load_state(state0)
xs_d, eps_d = xs.to(dev), eps.to(dev)
orc = None if args.no_oracle else M.StepOracle(spec, state0)
for s in range(args.steps):
    i = s % args.n_data
    curv = s >= args.curv_after
    eng.train_step(xs_d[i], eps_d[i], 1.0, curv)
    if orc is not None:
        out = orc.train_step(xs[i], eps[i], 1.0, epoch=12 if curv else 9)
    if (s + 1) % args.every == 0 or s == args.steps - 1:
        st = eng.read_stats()["last"]
        pv = eng.param_views()
        radii = [round(float(v), 4) for k, v in pv.items() if k.endswith("radius")]
        line = f"step {s+1:6d}  hip elbo/sample {st['elbo']/128:10.4f}  radii {radii}"
        if orc is not None:
            worst = max(float((pv[n].cpu() - orc.P[n].detach()).abs().max() / orc.P[n].detach().abs().max().clamp_min(1e-30))
                        for n in pv)
            oradii = [round(float(v), 4) for k, v in orc.P.items() if k.endswith("radius")]
            line += f" | oracle elbo/sample {float(out.elbo)/128:10.4f} radii {oradii}"
        print(line, flush=True)
