"""Dev probe: error of the HIP inverse maps against the reference's float32 goldens, next to the reference's own float32-vs-
float64 deviation (the conditioning of the point), for every (manifold, R, d) of g1_primitives."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import load_npz
from mvae_amd import functional as Fn
dev = torch.device("cuda:0")
g = load_npz("g1_primitives.npz")
T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
worst = {}
for man, kind in (("H", 1), ("S", 2), ("E", 0)):
    for R in (0.5, 1.0, 2.0, 11.0):
        for d in (2, 5, 40):
            k = f"{man}/R{R:g}/d{d}/f32/"
            k64 = f"{man}/R{R:g}/d{d}/f64/"
            Rt = torch.tensor(R, device=dev)
            mu, z = T(g[k + "mu"]).to(dev), T(g[k + "z"]).to(dev)
            iu, iv = Fn.inverse_sample_projection_mu0(kind, z, mu, Rt)
            lm = Fn.inverse_exp_map_mu0(kind, mu, Rt)
            outs = {"inv_u": iu, "inv_v": iv, "log_mu0": lm}
            if man != "E":
                mu0 = torch.zeros_like(mu); mu0[..., 0] = R
                a, b = Fn.inverse_sample_projection_mu0(kind, z, mu0, Rt)
                outs["inv0_u"], outs["inv0_v"] = a, b
            for nm, t in outs.items():
                ref = g[k + nm].astype(np.float64)
                sc = max(np.abs(ref).max(), 1e-30)
                e = np.abs(t.cpu().numpy().astype(np.float64) - ref).max() / sc
                r = np.abs(ref - g[k64 + nm]).max() / sc
                key = (nm, man)
                if key not in worst or e > worst[key][0]:
                    worst[key] = (e, r, R, d)
for (nm, man), (e, r, R, d) in sorted(worst.items()):
    print(f"{nm:8s} {man}  worst HIP-vs-ref(f32) {e:.2e}  (reference f32-vs-f64 at that point {r:.2e})  at R={R:g} d={d}")
