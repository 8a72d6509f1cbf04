"""Dev probe (GPU): parameters after 5 optimizer steps of the B=4 conv golden, in contraction modes 0 / 1 / 2, fused and
generic latent section: distance of the (sum, L2, max) summaries to the reference's record, and entry-wise differences
between the modes.  Answers whether a summary miss is a kernel error or Adam's sign noise on near-zero gradients."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import load_json, load_npz, summary_of
from mvae_amd import synthetic
from mvae_amd._lib import load
from mvae_amd.conv import ConvEngine
from oracle import model as M

dev = torch.device("cuda:0")
g = load_npz("g3_step_full.npz")
meta = load_json("g3_step_full.json")["cifar_conv_h2s2e2_learn"]
spec = M.Spec(meta["model"], in_dim=3072, h_dim=8192, arch="conv", fixed_curvature=False)
state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0, transposed_conv=("d1", "d2", "d3"))
B, steps = meta["batch"], 5
key = f"cifar_conv_h2s2e2_learn/f32/steps{steps}/"
xs = synthetic.uniform_batches(steps, B, 3072).to(dev)
eps = synthetic.eps_batches(steps, B, spec.total_true_dim).to(dev)
runs = {}
for fused in ("1", "0"):
    os.environ["MVAE_CONV_FUSED"] = fused
    for mode in (0, 1, 2):
        load().mvae_set_contraction_mode(mode)
        eng = ConvEngine([(c.letter, c.true_dim) for c in spec.components], dev, radius_trainable=[True] * 3)
        eng.load_state(state0)
        grads1 = None
        for s in range(steps):
            eng.forward_backward(xs[s], eps[s], 1.0)
            if s == 0:
                grads1 = {n: t.detach().cpu().numpy().copy() for n, t in eng.grad_views().items()}
            eng.optimizer_step(True)
        torch.cuda.synchronize()
        P = {n: t.detach().cpu().numpy().copy() for n, t in eng.param_views().items()}
        runs[(fused, mode)] = (P, grads1)
        worst = ("", 0.0)
        for n, t in P.items():
            ref = g[key + "state_final_summary/" + n]
            got = summary_of(t, ref)
            d = np.abs(got[:3] - ref[:3])
            lim = 1e-3 * steps * 1.01 + 5e-4 * np.abs(ref[:3]).max()
            k = (len(ref) - 3) // 2
            ds = np.abs(got[3 + k:] - ref[3 + k:]).max()
            if d.max() / lim > worst[1]:
                worst = (n, d.max() / lim)
            if d.max() > 0.5 * lim:
                print(f"  fused {fused} mode {mode} {n:34s} |d(sum,L2,max)| = {d[0]:.2e} {d[1]:.2e} {d[2]:.2e}  limit {lim:.2e}  "
                      f"sampled entries max diff {ds:.2e}  size {t.size}")
        print(f"fused {fused} mode {mode}: worst summary distance / limit = {worst[1]:.2f} ({worst[0]})")
load().mvae_set_contraction_mode(2)
for fused in ("1", "0"):
    P0, G0 = runs[(fused, 0)]
    for mode in (1, 2):
        P, G = runs[(fused, mode)]
        for n in P0:
            dp = np.abs(P[n] - P0[n])
            dg = np.abs(G[n] - G0[n])
            sc = max(np.abs(G0[n]).max(), 1e-30)
            big = int((dp > 1e-4).sum())
            if big or dg.max() / sc > 1e-4:
                print(f"fused {fused} mode {mode} vs 0  {n:34s} step-1 grad max diff / scale {dg.max() / sc:.2e}; params after 5: "
                      f"{big}/{dp.size} entries differ by > 1e-4 (max {dp.max():.2e}, sum of diffs {float((P[n] - P0[n]).sum()):.2e})")

# ---- lockstep: generic latent section, mode 2 next to mode 0, ReLU-output sign changes and gradient distance at every step
print("lockstep, fused 0: mode 2 next to mode 0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helpers import relu_flips, rel_l2
os.environ["MVAE_CONV_FUSED"] = "0"
engs = {}
for mode in (0, 2):
    load().mvae_set_contraction_mode(mode)
    e = ConvEngine([(c.letter, c.true_dim) for c in spec.components], dev, radius_trainable=[True] * 3)
    e.load_state(state0)
    engs[mode] = e
for s in range(steps):
    acts = {}
    for mode in (0, 2):
        load().mvae_set_contraction_mode(mode)
        acts[mode] = {k: v.clone() for k, v in engs[mode]._forward(xs[s], eps[s]).items() if k in ("a0", "a1", "a2", "t0", "b1", "b2")}
        engs[mode].forward_backward(xs[s], eps[s], 1.0)
    fl = relu_flips(acts[2], acts[0])
    gd = max(rel_l2(a.cpu().numpy(), b.cpu().numpy()) for (n, a), (_, b) in zip(engs[2].grad_views().items(), engs[0].grad_views().items()))
    pd = max(float((a - b).abs().max()) for (n, a), (_, b) in zip(engs[2].param_views().items(), engs[0].param_views().items()))
    print(f" step {s}: ReLU outputs with another sign {fl}; worst gradient rel-L2 {gd:.2e}; params before the step differ by <= {pd:.2e}")
    for mode in (0, 2):
        load().mvae_set_contraction_mode(mode)
        engs[mode].optimizer_step(True)
load().mvae_set_contraction_mode(2)
