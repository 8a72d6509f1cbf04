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
lib.mvae_debug_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for i in range(20): eng.train_step(xs[i % 8], eps[i % 8], 1.0, False)
torch.cuda.synchronize()
acc = None
N = 50
for i in range(N):
    eng.train_step(xs[i % 8], eps[i % 8], 1.0, False)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    lib.mvae_debug_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.mvae_debug_read(buf, 32)
    v = [int(x) for x in buf]
    d = [(v[1]-v[0]), (v[2]-v[1]), (v[3]-v[2]), (v[4]-v[3]), (v[9]-v[8]), (v[10]-v[9]), (v[11]-v[10]), (v[12]-v[11]),
         (v[17]-v[16]), (v[18]-v[17])]
    sp = (C.c_ulonglong * (3 * 2048))()
    lib.mvae_debug_read_spans(sp)
    st, en, kd = sp[0:2048], sp[2048:4096], sp[4096:6144]
    t0 = min(t for t, k in zip(st, kd) if k)
    for kind in range(24, 31):
        ends = [e - t0 for e, k in zip(en, kd) if k == kind]
        starts = [t - t0 for t, k in zip(st, kd) if k == kind]
        d.append(max(ends) if ends else 0)
        d.append(max(starts) if starts else 0)
    acc = d if acc is None else [a + b for a, b in zip(acc, d)]
names = ["fwd:load+sync", "fwd:heads", "fwd:comps", "fwd:dec0", "bwd:load+sync", "bwd:dz", "bwd:dot", "bwd:dh",
         "enc_bwd tile:loads+mfma", "enc_bwd tile:adam+stores", "dW_e0 last end", "dW_e0 last start", "dW_heads last end", "dW_heads last start",
         "dW_d0 last end", "dW_d0 last start", "b_e0 last end", "b_e0 last start", "b_heads last end",
         "b_heads last start", "b_d0 last end", "b_d0 last start", "radii end", "radii start"]
for n, a in zip(names, acc): print(f"{n:16s} {a / N * 10:8.1f} ns")
# distribution over the dW_e0 workgroups of the LAST step (10 ns ticks)
import numpy as np
ends = np.array(sorted((e - t0) * 10 for e, k in zip(en, kd) if k == 24))
print("dW_e0 workgroup end times (ns): min %d  p10 %d  p50 %d  p90 %d  p99 %d  max %d" %
      (ends[0], *np.percentile(ends, [10, 50, 90, 99]), ends[-1]))
idx = [i for i, k in enumerate(kd) if k == 24]
late = sorted(idx, key=lambda i: en[i])[-12:]
print("latest dW_e0 workgroups (block, end ns):", [(i, (en[i] - t0) * 10) for i in late])
