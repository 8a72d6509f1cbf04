"""Dev tool: phase timestamps of the latent kernels (needs a -DMV_DBG_TIMING build passed via MVAE_HIP_LIB)."""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import _lib, synthetic
from mvae_amd.engine import StepEngine
lib = _lib.load()
dev = torch.device("cuda:0")
MODEL = sys.argv[1] if len(sys.argv) > 1 else "h2,s2,e2"
from mvae_amd.utils import parse_component_str
comps = []
for tok in MODEL.split(","):
    mult, letter, dim = parse_component_str(tok)
    comps += [(letter, dim)] * mult
eng = StepEngine(comps, 784, 400, dev, radius_trainable=[l != "e" for l, _ in comps])
eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
xs = synthetic.binary_batches(8, 128, 784).to(dev); eps = synthetic.eps_batches(8, 128, eng.layout.eps_dim).to(dev)
for i in range(20): eng.train_step(xs[i % 8], eps[i % 8], 1.0, False)
torch.cuda.synchronize()
acc = None
N = 50
for i in range(N):
    eng.train_step(xs[i % 8], eps[i % 8], 1.0, False)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.mvae_debug_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.mvae_debug_read(buf, 16)
    v = [int(x) for x in buf]
    d = [(v[1]-v[0]), (v[2]-v[1]), (v[3]-v[2]), (v[4]-v[3]), (v[9]-v[8]), (v[10]-v[9]), (v[11]-v[10]), (v[12]-v[11]),
         (v[6]-v[5]), (v[7]-v[6])]
    acc = d if acc is None else [a + b for a, b in zip(acc, d)]
names = ["fwd:load+sync", "fwd:heads", "fwd:comps", "fwd:dec0", "bwd:load+sync", "bwd:dz", "bwd:dot", "bwd:dh",
         "side:tables", "side:duals"]
for n, a in zip(names, acc): print(f"{n:16s} {a / N * 10:8.1f} ns")
