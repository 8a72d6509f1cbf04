"""Dev tool: the fused step must be bit-reproducible.  Runs the same steps twice from the same state (eager launches) and
once as a HIP graph, and reports the first step / parameter segment whose bits differ."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import synthetic
from mvae_amd.engine import StepEngine

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def fresh():
    e = StepEngine([("h", 2), ("s", 2), ("e", 2)], 784, 400, dev, radius_trainable=[True, True, False])
    e.load_state(synthetic.synthetic_state([(n, s) for n, _, s in e.flat.entries], radius=2.0))
    return e


eng = fresh()
xs = synthetic.binary_batches(8, B, 784).to(dev)
eps = synthetic.eps_batches(8, B, eng.layout.eps_dim).to(dev)


def run(n, graph=False):
    global eng
    eng = fresh()  # parameters, Adam moments, step counter, statistics: all as at construction
    out = []
    if graph:
        keep = [t.clone() for t in (eng.params, eng.adam_m, eng.adam_v, eng.counters, eng.stats)]
        for i in range(2):
            eng.train_step(xs[i % 8], eps[i % 8], 1.0, True)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(n):
                eng.train_step(xs[i % 8], eps[i % 8], 1.0, True)
        torch.cuda.synchronize()
        for dst, src in zip((eng.params, eng.adam_m, eng.adam_v, eng.counters, eng.stats), keep):
            dst.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        return [(eng.params.clone(), eng.grads.clone())]
    for i in range(n):
        eng.train_step(xs[i % 8], eps[i % 8], 1.0, True)
        torch.cuda.synchronize()
        out.append((eng.params.clone(), eng.grads.clone()))
    return out


def segs(a, b):
    bad = []
    for name, off, shape in eng.flat.entries:
        n = 1
        for d in shape:
            n *= d
        if not torch.equal(a[off:off + n], b[off:off + n]):
            d = (a[off:off + n] - b[off:off + n]).abs()
            bad.append(f"{name}: {int((d > 0).sum())} of {n} differ, max {float(d.max()):.3e}")
    return bad


N = 6
r1, r2 = run(N), run(N)
for k in range(N):
    for what, idx in (("grads", 1), ("params", 0)):
        bad = segs(r1[k][idx], r2[k][idx])
        if bad:
            print(f"eager vs eager: step {k + 1} {what} differ:", bad)
            break
rg = run(N, graph=True)
bad = segs(rg[0][0], r1[-1][0])
print("graph vs eager after", N, "steps:", bad or "bit-identical")
print("eager twice:", "bit-identical" if not any(segs(r1[k][0], r2[k][0]) for k in range(N)) else "DIFFERENT")
