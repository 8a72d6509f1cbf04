"""Dev tool: us/step of the fused step for a model string at several batch sizes: python tools/bench_batch_model.py 6h2,6s2,6e2 100 112 128"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import synthetic
from mvae_amd.engine import StepEngine
from mvae_amd.runner import StepRunner
from mvae_amd.utils import parse_component_str
dev = torch.device("cuda:0")
comps = []
for tok in sys.argv[1].split(","):
    mult, letter, dim = parse_component_str(tok)
    comps += [(letter, dim)] * mult
for B in [int(a) for a in sys.argv[2:]]:
    eng = StepEngine(comps, 784, 400, dev, radius_trainable=[l != "e" for l, _ in comps])
    eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
    xs = synthetic.digits_like_batches(100, B).to(dev)
    eps = synthetic.eps_batches(100, B, eng.layout.eps_dim).to(dev)
    r = StepRunner(eng, xs, eps, beta=1.0, do_curvature_step=True, graph_steps=50, reset_every=2000)
    r.run(500)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.run(3000)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{sys.argv[1]} B={B:4d} path={eng.kernel_path(B):6s} {dt / 3000 * 1e6:7.1f} us/step  {3000 / dt:8.0f} steps/s")
