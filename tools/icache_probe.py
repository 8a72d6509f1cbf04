"""Dev probe: the same component kernel launched back-to-back (warm instruction cache) vs interleaved with other
kernels (cold).  Prints average kernel-to-kernel time from HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import functional as Fn
dev = torch.device("cuda:0")
lay = Fn.ComponentLayout([("h", 2), ("s", 2), ("e", 2)])
g = torch.Generator().manual_seed(0)
heads = (torch.randn(128, 12, generator=g) * 0.5).to(dev)
eps = torch.randn(128, 6, generator=g).to(dev)
radii = torch.tensor([2.0, 2.0, 0.0], device=dev)
dz = torch.randn(128, 8, generator=g).to(dev)
big = torch.randn(4096, 1024, device=dev)
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
f = lambda: Fn.component_forward(lay, heads, eps, radii)
b = lambda: Fn.component_backward(lay, heads, eps, radii, dz, None, 1.0)
print(f"component_forward  back-to-back: {timeit(f):7.2f} us per call (incl. host launch + allocs)")
print(f"component_backward back-to-back: {timeit(b):7.2f} us per call")
