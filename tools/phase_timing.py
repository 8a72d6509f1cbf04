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
    buf = (C.c_ulonglong * 48)()
    lib.mvae_debug_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.mvae_debug_read(buf, 48)
    v = [int(x) for x in buf]
    d = [(v[1]-v[0]), (v[2]-v[1]), (v[3]-v[2]), (v[4]-v[3]), (v[9]-v[8]), (v[10]-v[9]), (v[11]-v[10]), (v[12]-v[11]),
         (v[17]-v[16]), (v[18]-v[17]), (v[21]-v[20]), (v[22]-v[21]), (v[23]-v[22])] + \
        [(v[25 + j] - v[24 + j]) for j in range(7)]
    sp = (C.c_ulonglong * (6 * 3 * 2048))()
    lib.mvae_debug_read_spans(sp)
    spans = {}
    bounds = []
    for L in range(6):
        base = L * 3 * 2048
        st, en, kd = sp[base:base + 2048], sp[base + 2048:base + 4096], sp[base + 4096:base + 6144]
        used = [i for i in range(2048) if kd[i]]
        if not used:  # launch not part of this step (launch 2 of the fused forward)
            continue
        t0 = min(st[i] for i in used)
        bounds.append((t0, max(en[i] for i in used)))
        if L == 5 and i == N - 1:  # the latest tile workgroups of launch 6 (which ones are the tail?)
            tiles = [j for j in used if kd[j] == 1]
            if tiles:
                j0 = min(tiles)
                late = sorted(tiles, key=lambda j: -en[j])[:12]
                print("launch 6: latest tile workgroups (index - first tile index, end ns):",
                      [(j - j0, (en[j] - t0) * 10) for j in late])
        for kind in sorted(set(kd[i] for i in used)):
            ends = sorted((en[i] - t0) * 10 for i in used if kd[i] == kind)
            starts = sorted((st[i] - t0) * 10 for i in used if kd[i] == kind)
            durs = sorted((en[i] - st[i]) * 10 for i in used if kd[i] == kind)
            spans.setdefault((L, kind), []).append((ends[len(ends) // 2], ends[-1], len(ends), starts[len(starts) // 2],
                                                    starts[-1], durs[len(durs) // 2], durs[-1]))
    nb = len(bounds)
    gaps = [(bounds[L + 1][0] - bounds[L][1]) * 10 for L in range(nb - 1)] + [(bounds[-1][1] - bounds[0][0]) * 10]
    gap_acc = gaps if i == 0 else [a + b for a, b in zip(gap_acc, gaps)]
    acc = d if acc is None else [a + b for a, b in zip(acc, d)]
    wv = [v[32 + j] for j in range(8)]
    wacc = wv if i == 0 else [a + b for a, b in zip(wacc, wv)]
    k56 = [v[41 + j] - v[40 + j] for j in range(6)]
    k56acc = k56 if i == 0 else [a + b for a, b in zip(k56acc, k56)]
names = ["fwd:load+sync", "fwd:heads", "fwd:comps", "fwd:dec0", "bwd:load+sync", "bwd:dz", "bwd:dot", "bwd:dh",
         "enc_bwd tile:loads+mfma", "enc_bwd tile:adam+stores", "dec1_fwd tile:loads+mfma", "dec1_fwd tile:reduce",
         "dec1_fwd tile:epilogue", "fwd23: loads+heads mfma", "fwd23: heads reduce", "fwd23: heads_s/tables",
         "fwd23: components", "fwd23: hd", "fwd23: logits mfma", "fwd23: reduce"]
for n, a in zip(names, acc): print(f"{n:16s} {a / N * 10:8.1f} ns")
print("fwd23 per-wave time to the end of the heads MFMA (ns):", [round(a / N * 10) for a in wacc])
print("k_bwd56 tile workgroup, wave 0 (ns): requests issued, dheads (waits for dz + records), dh fragments, meet, tile MFMAs, "
      "Adam + stores:", [round(a / N * 10) for a in k56acc])
KERNELS = ["enc_fwd", "latent_fwd", "dec1_fwd", "dec1_bwd", "latent_bwd", "enc_bwd"]
KINDS = {(0, 1): "tiles", (1, 1): "main waves", (1, 2): "dual waves", (2, 1): "tiles", (2, 2): "dual workgroups", (3, 1): "dhd tiles",
         (3, 2): "db_logits", (3, 3): "statistics", (3, 4): "dual records", (4, 1): "rows", (4, 2): "dW_logits tiles", (4, 3): "statistics", (5, 1): "dW_e0 tiles",
         (5, 2): "dW_logits waves (k_bwd56) | dW_heads", (5, 3): "dW_d0", (5, 4): "b_e0", (5, 5): "b_heads", (5, 6): "b_d0", (5, 7): "radii"}
import numpy as np
print("per launch: workgroup END time after the first workgroup's start, ns (median / latest over workgroups)")
for (L, kind), v in sorted(spans.items()):
    med = np.mean([x[0] for x in v]); mx = np.mean([x[1] for x in v])
    sm_, sx = np.mean([x[3] for x in v]), np.mean([x[4] for x in v])
    dm, dx = np.mean([x[5] for x in v]), np.mean([x[6] for x in v])
    print(f"  {KERNELS[L]:11s} {KINDS.get((L, kind), kind):16s} n={v[0][2]:4d}  end median {med:6.0f} latest {mx:6.0f} | "
          f"start median {sm_:5.0f} latest {sx:5.0f} | duration median {dm:5.0f} longest {dx:5.0f}")
print("idle between the last workgroup (thread 0) of a launch and the first workgroup of the next, ns:",
      [round(g / N) for g in gap_acc[:-1]], " first start -> last end of the step:", round(gap_acc[-1] / N))
