"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (oskopek/mvae).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference is imported through `ref_shim.py` (stubs for absent import-time deps, no arithmetic touched).
Only inputs and outputs are stored -- no reference source text.  Every array is produced by the reference's own
functions; this script only prepares seeded inputs, calls them, and records what comes back.

Files written (all small):
  g1_primitives.npz     manifold primitives H/S/E  x  R in {0.5,1,2,11}  x  d in {2,5,40}, f32 and f64
  g1_scalar_fns.npz     guarded scalar functions + their (custom) gradients, f32 and f64
  g2_component.npz      Component.encode -> reparametrize -> rsample_with_parts -> kl_loss, with gradients
  g3_step_small.npz     full train_step (fwd, ELBO, bwd, CurvatureOptimizer) on small models, all tensors
  g3_step_full.npz      same at the benchmark size (B=128, h=400, D=784): per-sample stats + summaries
  g4_loglik.npz         ModelVAE.log_likelihood(x, n=8)
  g5_parser.json        model-string grammar table
  g7_distances.npz      geodesic distances through the helpers of the reference's own op tests (lorentz_distance,
                        spherical_distance, euclidean_distance) and the lorentz product / norm, H/S/E x R x d, f32 and f64
  g6_projected.npz      the reference-owned part of the projected sphere `d` (everything that does not cross into
                        geoopt's mobius_add) and Universal.radius / _choice, f32 and f64
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root (for mvae_amd.synthetic)

import ref_shim  # noqa: E402

ref_shim.install()

from mt.mvae import utils as ref_utils  # noqa: E402
from mt.mvae.ops import common as C  # noqa: E402
from mt.mvae.ops import hyperbolics as H  # noqa: E402
from mt.mvae.ops import spherical as S  # noqa: E402
from mt.mvae.ops import euclidean as E  # noqa: E402
from mt.mvae.ops import Hyperboloid, Sphere, Euclidean  # noqa: E402
from mt.mvae.models import FeedForwardVAE, ConvolutionalVAE, Trainer  # noqa: E402
from mt.data import VaeDataset  # noqa: E402

from mvae_amd import synthetic  # noqa: E402

DTYPES = {"f32": torch.float32, "f64": torch.float64}


def npy(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------------------------- G6
def gen_projected():
    from mt.mvae.ops import spherical_projected as SP
    from mt.mvae.ops import StereographicallyProjectedSphere, Universal
    out = {}
    for dname, dtype in DTYPES.items():
        torch.set_default_dtype(dtype)
        for R in [0.5, 1.0, 2.0, 11.0]:
            for d in [2, 5, 40]:
                rows = 8
                g = torch.Generator().manual_seed(7000 + int(R * 100) + d)
                x = (torch.randn(rows, d, generator=g, dtype=torch.float64) * 0.6 * min(R, 3.0) / np.sqrt(d)).to(dtype)
                y = (torch.randn(rows, d, generator=g, dtype=torch.float64) * 0.8 * R / np.sqrt(d)).to(dtype)
                w = (torch.randn(rows, d, generator=g, dtype=torch.float64) * 0.8 * R / np.sqrt(d)).to(dtype)
                radius = torch.tensor(R, dtype=dtype)
                key = f"D/R{R:g}/d{d}/{dname}/"
                out[key + "x"], out[key + "y"], out[key + "w"] = npy(x), npy(y), npy(w)
                mu = SP.exp_map_mu0(x, radius=radius)
                out[key + "mu"] = npy(mu)
                out[key + "log_mu0"] = npy(SP.inverse_exp_map_mu0(mu, radius=radius))
                out[key + "pt"] = npy(SP.parallel_transport_mu0(x, dst=y, radius=radius))
                out[key + "ipt"] = npy(SP.inverse_parallel_transport_mu0(x, src=y, radius=radius))
                out[key + "lambda"] = npy(SP.lambda_x(y, radius))
                out[key + "to_sphere"] = npy(SP.projected_to_spherical(y, radius))
                out[key + "dist"] = npy(SP.spherical_projected_distance(y, w, SP._c(radius)))
                man = StereographicallyProjectedSphere(lambda: radius)
                out[key + "logdet"] = npy(man.logdet(y, None, w, ()))  # (mu, std, z, data): uses mu and z only
                out[key + "logdet0"] = npy(man.logdet(SP.mu_0(w.shape), None, w, ()))
        ks = torch.tensor([-4.0, -0.25, -1e-5, -1e-7, 0.0, 1e-7, 1e-5, 0.25, 4.0], dtype=dtype)
        out[f"U/{dname}/K"] = npy(ks)
        out[f"U/{dname}/radius"] = npy(torch.stack([Universal(lambda k=k: k).radius for k in ks]))
        out[f"U/{dname}/choice"] = np.array([Universal(lambda k=k: k)._choice for k in ks], dtype=np.int64)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g6_projected.npz"), **out)
    print("g6_projected:", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- G7
def _load_reference_test_module(name):
    """Imports tests/mvae/ops/<name>.py of the REFERENCE by path (its helpers hold the distance formulas)."""
    import importlib.util
    path = os.path.join(ref_shim.REFERENCE_ROOT, "tests", "mvae", "ops", name + ".py")
    spec = importlib.util.spec_from_file_location("ref_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_distances():
    out = {}
    for dname, dtype in DTYPES.items():
        torch.set_default_dtype(dtype)
        TH = _load_reference_test_module("test_hyperbolics")
        TS = _load_reference_test_module("test_spherical")
        TE = _load_reference_test_module("test_euclidean")
        for mname, M in {"H": H, "S": S, "E": E}.items():
            for R in [0.5, 1.0, 2.0, 11.0]:
                for d in [2, 5, 40]:
                    rows = 8
                    g = torch.Generator().manual_seed(9000 + int(R * 100) + d)
                    x = (torch.randn(rows, d, generator=g, dtype=torch.float64) * 0.6 * min(R, 3.0) / np.sqrt(d)).to(dtype)
                    y = (torch.randn(rows, d, generator=g, dtype=torch.float64) * 0.6 * min(R, 3.0) / np.sqrt(d)).to(dtype)
                    radius = torch.tensor(R, dtype=dtype)
                    key = f"{mname}/R{R:g}/d{d}/{dname}/"
                    out[key + "x"], out[key + "y"] = npy(x), npy(y)
                    if mname == "E":
                        p, q = M.exp_map_mu0(x), M.exp_map_mu0(y)
                        out[key + "p"], out[key + "q"] = npy(p), npy(q)
                        out[key + "dist"] = npy(TE.euclidean_distance(p, q))
                        continue
                    p = M.exp_map_mu0(C.expand_proj_dims(x), radius=radius)
                    q = M.exp_map_mu0(C.expand_proj_dims(y), radius=radius)
                    out[key + "p"], out[key + "q"] = npy(p), npy(q)
                    if mname == "H":
                        TH.radius = radius  # the helper reads the module-level radius of its test file
                        out[key + "dist"] = npy(TH.lorentz_distance(p, q, keepdim=True))
                        out[key + "lprod"] = npy(H.lorentz_product(p, q, keepdim=True))
                        out[key + "lnorm_u"] = npy(H.lorentz_norm(H.inverse_exp_map(q, at_point=p, radius=radius),
                                                                  keepdim=True))
                    else:
                        out[key + "dist"] = npy(TS.spherical_distance(p, q, radius))
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g7_distances.npz"), **out)
    print("g7_distances:", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- G1
def gen_primitives():
    out = {}
    mods = {"H": H, "S": S, "E": E}
    for dname, dtype in DTYPES.items():
        torch.set_default_dtype(dtype)
        for mname, M in mods.items():
            for R in [0.5, 1.0, 2.0, 11.0]:
                for d in [2, 5, 40]:
                    rows = 8
                    g = torch.Generator().manual_seed(int(R * 100) + d)
                    # tangent vector at mu0 of moderate geodesic length (scaled with R), a "noise" vector v
                    x = (torch.randn(rows, d, generator=g, dtype=torch.float64) * 0.6 * min(R, 3.0) /
                         np.sqrt(d)).to(dtype)
                    v = (torch.randn(rows, d, generator=g, dtype=torch.float64) * 0.5 * min(R, 3.0) /
                         np.sqrt(d)).to(dtype)
                    radius = torch.tensor(R, dtype=dtype)
                    key = f"{mname}/R{R:g}/d{d}/{dname}/"
                    out[key + "x"] = npy(x)
                    out[key + "v"] = npy(v)
                    if mname == "E":
                        mu = M.exp_map_mu0(x)
                        z, (u, v_) = M.sample_projection_mu0(v, mu)
                        u2, v2 = M.inverse_sample_projection_mu0(z, mu)
                        out[key + "mu"] = npy(mu)
                        out[key + "z"] = npy(z)
                        out[key + "u"] = npy(u)
                        out[key + "inv_u"] = npy(u2)
                        out[key + "inv_v"] = npy(v2)
                        out[key + "log_mu0"] = npy(M.inverse_exp_map_mu0(mu))
                        continue
                    mu = M.exp_map_mu0(C.expand_proj_dims(x), radius=radius)
                    z, (u, v_) = M.sample_projection_mu0(v, mu, radius=radius)
                    u2, v2 = M.inverse_sample_projection_mu0(z, mu, radius=radius)
                    mu0 = M.mu_0(mu.shape, radius=radius)
                    u0, v0 = M.inverse_sample_projection_mu0(z, mu0, radius=radius)
                    out[key + "mu"] = npy(mu)
                    out[key + "z"] = npy(z)
                    out[key + "u"] = npy(u)
                    out[key + "inv_u"] = npy(u2)
                    out[key + "inv_v"] = npy(v2)
                    out[key + "inv0_u"] = npy(u0)
                    out[key + "inv0_v"] = npy(v0)
                    out[key + "logdet_u"] = npy(M._logdet(u, radius))
                    out[key + "logdet_u0"] = npy(M._logdet(u0, radius))
                    out[key + "pt"] = npy(M.parallel_transport_mu0(C.expand_proj_dims(v), dst=mu, radius=radius))
                    out[key + "ipt"] = npy(M.inverse_parallel_transport_mu0(u, src=mu, radius=radius))
                    out[key + "exp"] = npy(M.exp_map(u, at_point=mu, radius=radius))
                    out[key + "log"] = npy(M.inverse_exp_map(z, at_point=mu, radius=radius))
                    out[key + "log_mu0"] = npy(M.inverse_exp_map_mu0(mu, radius=radius))
                    if mname == "H":
                        out[key + "to_poincare"] = npy(H.lorentz_to_poincare(mu, radius=radius))
                    else:
                        out[key + "to_projected"] = npy(S.spherical_to_projected(mu, radius=radius))
    np.savez_compressed(os.path.join(HERE, "g1_primitives.npz"), **out)
    print("g1_primitives:", len(out), "arrays")


def gen_scalar_fns():
    out = {}
    fns = {
        "acosh": C.acosh, "atanh": C.atanh, "cosh": C.cosh, "sinh": C.sinh, "sqrt": C.sqrt, "logsinh": C.logsinh,
        "logcosh": C.logcosh, "clamp_m1_2": lambda t: C.clamp(t, min=-1.0, max=2.0),
    }
    pts = {
        "acosh": [0.0, 0.5, 1.0, 1.0 + 1e-7, 1.0 + 1e-4, 1.001, 1.5, 2.0, 10.0, 1e3, 1e6],
        "atanh": [-2.0, -1.0, -0.999999, -0.5, 0.0, 1e-4, 0.5, 0.9, 0.99999, 1.0, 3.0],
        "cosh": [-500.0, -85.0, -84.0, -3.0, 0.0, 1e-3, 2.0, 50.0, 85.0, 86.0, 500.0],
        "sinh": [-500.0, -85.0, -84.0, -3.0, 0.0, 1e-3, 2.0, 50.0, 85.0, 86.0, 500.0],
        "sqrt": [-1.0, 0.0, 1e-12, 1e-9, 2e-9, 1e-6, 0.25, 1.0, 2.0, 1e4, 1e10],
        "logsinh": [1e-6, 1e-4, 1e-3, 1e-2, 0.1, 0.5, 1.0, 3.0, 10.0, 50.0, 400.0],
        "logcosh": [-400.0, -10.0, -1.0, -1e-3, 0.0, 1e-3, 0.5, 1.0, 3.0, 50.0, 400.0],
        "clamp_m1_2": [-5.0, -1.0, -0.999, 0.0, 1.0, 1.999, 2.0, 2.001, 7.0, 1e3, -1e3],
    }
    for dname, dtype in DTYPES.items():
        torch.set_default_dtype(dtype)
        for name, fn in fns.items():
            x = torch.tensor(pts[name], dtype=dtype, requires_grad=True)
            y = fn(x)
            (gx,) = torch.autograd.grad(y.sum(), x)
            out[f"{name}/{dname}/x"] = npy(x)
            out[f"{name}/{dname}/y"] = npy(y)
            out[f"{name}/{dname}/dy"] = npy(gx)
    # acosh known-answer inputs of the reference's test_common.py:45-50 (np.random.seed(42) stream)
    np.random.seed(42)
    _ = (np.random.random_sample(100) - 0.5) * 1000.0
    xs = np.random.random_sample(100) * 100 + 1
    torch.set_default_dtype(torch.float32)
    out["acosh_known/x"] = xs
    out["acosh_known/y_f32"] = npy(C.acosh(torch.tensor(xs).float()))
    np.savez_compressed(os.path.join(HERE, "g1_scalar_fns.npz"), **out)
    print("g1_scalar_fns:", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- G2
def gen_component():
    """Component-level: (mean_raw, logvar_raw, eps, R) -> z, kl (+ parts) and gradients of a fixed linear loss."""
    out = {}
    B = 16
    for dname, dtype in DTYPES.items():
        torch.set_default_dtype(dtype)
        for letter in ["h", "s", "e"]:
            for R in [1.0, 2.0, 11.0]:
                for d in [2, 5]:
                    for scalar in [False, True]:
                        if letter == "e" and R != 1.0:
                            continue
                        comp = ref_utils.parse_components(f"{letter}{d}", fixed_curvature=False)[0]
                        comp.init_layers(4, scalar_parametrization=scalar)
                        for n in ("_nradius", "_pradius"):
                            if hasattr(comp, n):
                                getattr(comp, n).data = torch.tensor(R, dtype=dtype)
                        g = torch.Generator().manual_seed(int(R * 10) + d + (100 if scalar else 0) + ord(letter))
                        mean_raw = (torch.randn(B, d, generator=g, dtype=torch.float64) * 0.7).to(dtype)
                        lv_raw = (torch.randn(B, 1 if scalar else d, generator=g, dtype=torch.float64) * 0.8).to(dtype)
                        wz = torch.randn(B, comp.dim, generator=g, dtype=torch.float64).to(dtype)
                        wkl = torch.rand(B, generator=g, dtype=torch.float64).to(dtype) + 0.5
                        mean_raw.requires_grad_(True)
                        lv_raw.requires_grad_(True)
                        # --- the body of Component.encode (component.py:63-75) with the two Linear outputs given:
                        z_mean_h = comp.manifold.exp_map_mu0(mean_raw)
                        std = F.softplus(lv_raw) + 1e-5
                        ref_shim.reseed_eps(77 + d)
                        q_z, p_z = comp.reparametrize(z_mean_h, std)
                        z, data = q_z.rsample_with_parts()
                        kl = comp.kl_loss(q_z, p_z, z, data)
                        eps = ref_shim.eps_log[-1]
                        loss = (wz * z).sum() + (wkl * kl).sum()
                        params = [mean_raw, lv_raw]
                        rad = None
                        for n in ("_nradius", "_pradius"):
                            if hasattr(comp, n):
                                rad = getattr(comp, n)
                                params.append(rad)
                        grads = torch.autograd.grad(loss, params)
                        key = f"{letter}{d}/R{R:g}/{'scalar' if scalar else 'diag'}/{dname}/"
                        out[key + "mean_raw"] = npy(mean_raw)
                        out[key + "logvar_raw"] = npy(lv_raw)
                        out[key + "eps"] = npy(eps)
                        out[key + "wz"] = npy(wz)
                        out[key + "wkl"] = npy(wkl)
                        out[key + "mu"] = npy(z_mean_h)
                        out[key + "std"] = npy(std)
                        out[key + "z"] = npy(z)
                        out[key + "kl"] = npy(kl)
                        if data is not None:
                            out[key + "u"] = npy(data[0])
                            out[key + "v"] = npy(data[1])
                            out[key + "logq"] = npy(q_z.log_prob_from_parts(z, data))
                            out[key + "logp"] = npy(p_z.log_prob(z))
                        out[key + "d_mean_raw"] = npy(grads[0])
                        out[key + "d_logvar_raw"] = npy(grads[1])
                        if rad is not None:
                            out[key + "d_radius"] = npy(grads[2])
    np.savez_compressed(os.path.join(HERE, "g2_component.npz"), **out)
    print("g2_component:", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- G3
class _BceDataset(VaeDataset):

    def __init__(self, batch, in_dim):
        super().__init__(batch, in_dim=in_dim, img_dims=None)

    def reconstruction_loss(self, x_mb_, x_mb):
        return F.binary_cross_entropy_with_logits(x_mb_, x_mb, reduction="none")


class _Loader:
    """Just enough of a DataLoader for Trainer._train_epoch (train.py:185-220)."""

    def __init__(self, xs):
        self.xs = xs
        self.dataset = [None] * (xs.shape[0] * xs.shape[1])

    def __iter__(self):
        for i in range(self.xs.shape[0]):
            yield self.xs[i], torch.zeros(self.xs.shape[1], dtype=torch.long)


def _build_reference_model(model_str, arch, in_dim, h_dim, batch, fixed, scalar, radius, dtype):
    components = ref_utils.parse_components(model_str, fixed)
    ds = _BceDataset(batch, in_dim)
    if arch == "ff":
        model = FeedForwardVAE(h_dim, components, ds, scalar)
        tconv = ()
    else:
        model = ConvolutionalVAE(h_dim, components, ds, scalar)
        tconv = ("d1", "d2", "d3")
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    state = synthetic.synthetic_state(shapes, radius=radius, dtype=dtype, transposed_conv=tconv)
    model.load_state_dict(state)
    return model, state


def run_reference(model_str, arch, in_dim, h_dim, batch, steps, epoch, fixed, scalar, dtype, radius=2.0, lr=1e-3,
                  beta=1.0, soft_targets=False):
    """Runs Trainer._train_epoch of the reference for `steps` batches at trainer.epoch=`epoch`."""
    torch.set_default_dtype(dtype)
    model, state0 = _build_reference_model(model_str, arch, in_dim, h_dim, batch, fixed, scalar, radius, dtype)
    trainer = Trainer(model, img_dims=None, chkpt_dir="/tmp/golden_chkpt")
    opt = trainer.build_optimizer(learning_rate=lr, fixed_curvature=fixed)
    trainer.epoch = epoch
    true_dims = sum(c.true_dim for c in model.components)
    gen = synthetic.uniform_batches if soft_targets else synthetic.binary_batches
    xs = gen(steps, batch, in_dim, dtype=dtype)
    eps_all = synthetic.eps_batches(steps, batch, true_dims, dtype=dtype)  # the recipe the tests regenerate
    queue, off = [], 0
    for s_ in range(steps):
        off = 0
        for c in model.components:
            queue.append(eps_all[s_, :, off:off + c.true_dim].clone())
            off += c.true_dim
    ref_shim.preload_eps(queue)

    per_step = []
    orig_train_step = model.train_step

    def recording_train_step(optimizer, x_mb, beta):
        stats, (reps, concat_z, x_) = orig_train_step(optimizer, x_mb, beta)
        rec = {
            "bce": stats.bce, "kl": stats.kl, "elbo": stats.elbo, "component_kl": list(stats.component_kl),
        }
        if len(per_step) == 0:  # tensors of the first step (before any update touched the graph outputs)
            rec["logits"] = npy(x_)
            rec["concat_z"] = npy(concat_z)
            rec["bce_rows"] = npy(captured["bs"]._bce)
            rec["kl_rows"] = np.stack([npy(k) for k in captured["bs"]._component_kl])
            rec["grads"] = {
                k: (npy(p.grad) if p.grad is not None else None) for k, p in model.named_parameters()
            }
            rec["state_after"] = {k: npy(v).copy() for k, v in model.state_dict().items()}
        per_step.append(rec)
        return stats, (reps, concat_z, x_)

    model.train_step = recording_train_step
    captured = {}
    orig_cbs = model.compute_batch_stats

    def recording_cbs(*a, **k):  # the per-sample tensors of the step itself (before the optimizer moves the radii)
        captured["bs"] = orig_cbs(*a, **k)
        return captured["bs"]

    model.compute_batch_stats = recording_cbs
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        epoch_stats = trainer._train_epoch(opt, _Loader(xs), beta=beta)
    eps = ref_shim.eps_log
    assert len(eps) == steps * len(model.components)
    ncomp = len(model.components)
    eps_steps = torch.stack(
        [torch.cat([eps[s * ncomp + i] for i in range(ncomp)], dim=-1) for s in range(steps)])
    assert eps_steps.shape == (steps, batch, true_dims)
    assert torch.equal(eps_steps, eps_all)
    return {
        "state0": {k: npy(v) for k, v in state0.items()},
        "radius_at_epoch_start": None,
        "x": npy(xs),
        "eps": npy(eps_steps),
        "steps": per_step,
        "state_final": {k: npy(v) for k, v in model.state_dict().items()},
        "epoch_stats": epoch_stats.to_print(),
    }


def _summary(a, rng):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = rng.choice(a.size, size=min(16, a.size), replace=False)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum()), np.abs(a).max()], idx.astype(np.float64), a[idx]])


SMALL_CASES = [
    # name, model, in_dim, h_dim, B, fixed, scalar, epoch
    ("e6_fixed_ep0", "e6", 32, 16, 8, True, False, 0),
    ("h2s2e2_learn_ep0", "h2,s2,e2", 32, 16, 8, False, False, 0),
    ("h2s2e2_learn_ep12", "h2,s2,e2", 32, 16, 8, False, False, 12),
    ("h2s2e2_fixed_ep12", "h2,s2,e2", 32, 16, 8, True, False, 12),
    ("h2s2e2_scalar_ep12", "h2,s2,e2", 32, 16, 8, False, True, 12),
    ("h2s2e2_d784_ep12", "h2,s2,e2", 784, 16, 8, False, False, 12),
    ("prod36_learn_ep12", "6h2,6s2,6e2", 32, 16, 8, False, False, 12),
    ("h5s3e4_learn_ep5", "h5,s3,e4", 32, 16, 8, False, False, 5),
]


def gen_steps_small():
    out = {}
    for dname, dtype in DTYPES.items():
        for name, model_str, in_dim, h_dim, B, fixed, scalar, epoch in SMALL_CASES:
            if dname == "f64" and in_dim > 32:
                continue  # keeps the fixture small; f64 is covered by the in_dim=32 cases
            for steps in (1, 5):
                r = run_reference(model_str, "ff", in_dim, h_dim, B, steps, epoch, fixed, scalar, dtype)
                key = f"{name}/{dname}/steps{steps}/"
                if steps == 1:
                    for k, v in r["state0"].items():
                        out[key + "state0/" + k] = v
                    s0 = r["steps"][0]
                    out[key + "logits"] = s0["logits"]
                    out[key + "concat_z"] = s0["concat_z"]
                    out[key + "bce_rows"] = s0["bce_rows"]
                    out[key + "kl_rows"] = s0["kl_rows"]
                    for k, v in s0["grads"].items():
                        if v is not None:
                            out[key + "grad/" + k] = v
                    for k, v in s0["state_after"].items():
                        out[key + "state1/" + k] = v
                out[key + "x"] = r["x"].astype(np.uint8)
                out[key + "eps"] = r["eps"]
                out[key + "stats"] = np.array(
                    [[s["bce"], s["kl"], s["elbo"]] + s["component_kl"] for s in r["steps"]], dtype=np.float64)
                for k, v in r["state_final"].items():
                    out[key + "state_final/" + k] = v
    np.savez_compressed(os.path.join(HERE, "g3_step_small.npz"), **out)
    meta = {n: dict(model=m, in_dim=i, h_dim=h, batch=b, fixed_curvature=f, scalar_parametrization=s, epoch=e)
            for n, m, i, h, b, f, s, e in SMALL_CASES}
    with open(os.path.join(HERE, "g3_step_small.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("g3_step_small:", len(out), "arrays")


FULL_CASES = [
    # name, model, arch, in_dim, h_dim, B, fixed, epoch, soft
    ("mnist_e6_fixed", "e6", "ff", 784, 400, 128, True, 12, False),
    ("mnist_h2s2e2_learn", "h2,s2,e2", "ff", 784, 400, 128, False, 12, False),
    ("mnist_prod36_learn", "6h2,6s2,6e2", "ff", 784, 400, 128, False, 12, False),
    ("cifar_conv_h2s2e2_learn", "h2,s2,e2", "conv", 3072, 8192, 4, False, 12, True),
]


def gen_steps_full():
    """Benchmark-size cases: inputs are regenerated by recipe (mvae_amd/synthetic.py); per-sample stats are stored
    in full, big tensors as (sum, L2, max|.|, 16 sampled entries)."""
    out = {}
    meta = {}
    for name, model_str, arch, in_dim, h_dim, B, fixed, epoch, soft in FULL_CASES:
        for dname, dtype in DTYPES.items():
            rng = np.random.RandomState(7)
            for steps in (1, 5):
                r = run_reference(model_str, arch, in_dim, h_dim, B, steps, epoch, fixed, False, dtype,
                                  soft_targets=soft)
                key = f"{name}/{dname}/steps{steps}/"
                if steps == 1:
                    s0 = r["steps"][0]
                    out[key + "concat_z"] = s0["concat_z"]
                    out[key + "bce_rows"] = s0["bce_rows"]
                    out[key + "kl_rows"] = s0["kl_rows"]
                    out[key + "logits_summary"] = _summary(s0["logits"], rng)
                    for k, v in s0["grads"].items():
                        if v is not None:
                            out[key + "grad_summary/" + k] = _summary(v, rng)
                    for k, v in s0["state_after"].items():
                        out[key + "state1_summary/" + k] = _summary(v, rng)
                out[key + "stats"] = np.array(
                    [[s["bce"], s["kl"], s["elbo"]] + s["component_kl"] for s in r["steps"]], dtype=np.float64)
                for k, v in r["state_final"].items():
                    out[key + "state_final_summary/" + k] = _summary(v, rng)
        meta[name] = dict(model=model_str, arch=arch, in_dim=in_dim, h_dim=h_dim, batch=B, fixed_curvature=fixed,
                          epoch=epoch, soft_targets=soft)
        print("  full case done:", name, flush=True)
    np.savez_compressed(os.path.join(HERE, "g3_step_full.npz"), **out)
    with open(os.path.join(HERE, "g3_step_full.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("g3_step_full:", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- G8
def _summary_n(a, rng, n=64):
    """(sum, L2, max|.|, n sampled indices, the n entries)"""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = rng.choice(a.size, size=min(n, a.size), replace=False)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum()), np.abs(a).max()], idx.astype(np.float64), a[idx]])


G8_CASES = [
    # name, model, arch, in_dim, h_dim, B, fixed, epoch, soft
    # BASELINE configs[4] at its REAL batch size (the B = 4 case of g3_step_full pins the layers, this one the size)
    ("cifar_conv_h2s2e2_learn_b256", "h2,s2,e2", "conv", 3072, 8192, 256, False, 12, True),
    # the reference's own large-component models (tests/mvae/models/test_vae.py:212-249) at the benchmark size
    ("mnist_h40_learn", "h40", "ff", 784, 400, 128, False, 12, False),
    ("mnist_s40_learn", "s40", "ff", 784, 400, 128, False, 12, False),
]


def gen_full_size_extra():
    """One reference step (forward, ELBO, backward, optimizer) per case and dtype: per-sample statistics in full, every
    big tensor as (sum, L2, max|.|, 64 sampled entries).  float32 AND float64, so that a test can state the float32
    reference's own distance from the float64 reference next to the HIP path's."""
    out, meta = {}, {}
    for name, model_str, arch, in_dim, h_dim, B, fixed, epoch, soft in G8_CASES:
        for dname, dtype in DTYPES.items():
            rng = np.random.RandomState(11)
            r = run_reference(model_str, arch, in_dim, h_dim, B, 1, epoch, fixed, False, dtype, soft_targets=soft)
            key = f"{name}/{dname}/"
            s0 = r["steps"][0]
            out[key + "concat_z"] = s0["concat_z"]
            out[key + "bce_rows"] = s0["bce_rows"]
            out[key + "kl_rows"] = s0["kl_rows"]
            out[key + "logits_summary"] = _summary_n(s0["logits"], rng)
            for k, v in s0["grads"].items():
                if v is not None:
                    out[key + "grad_summary/" + k] = _summary_n(v, rng)
            for k, v in s0["state_after"].items():
                out[key + "state1_summary/" + k] = _summary_n(v, rng)
            out[key + "stats"] = np.array([[s["bce"], s["kl"], s["elbo"]] + s["component_kl"] for s in r["steps"]],
                                          dtype=np.float64)
        meta[name] = dict(model=model_str, arch=arch, in_dim=in_dim, h_dim=h_dim, batch=B, fixed_curvature=fixed,
                          epoch=epoch, soft_targets=soft)
        print("  g8 case done:", name, flush=True)
    np.savez_compressed(os.path.join(HERE, "g8_full_size_extra.npz"), **out)
    with open(os.path.join(HERE, "g8_full_size_extra.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("g8_full_size_extra:", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- G4
def gen_loglik():
    out = {}
    for dname, dtype in DTYPES.items():
        torch.set_default_dtype(dtype)
        for name, model_str in [("h2s2e2", "h2,s2,e2"), ("e6", "e6"), ("h5s3e4", "h5,s3,e4")]:
            B, in_dim, h_dim, n = 8, 32, 16, 8
            model, state0 = _build_reference_model(model_str, "ff", in_dim, h_dim, B, False, False, 2.0, dtype)
            x = synthetic.binary_batches(1, B, in_dim, dtype=dtype)[0]
            ref_shim.reseed_eps(2000)
            with torch.no_grad():
                log_px, mi, cov_norm = model.log_likelihood(x, n=n)
            eps = torch.cat(list(ref_shim.eps_log), dim=-1)  # [n, B, sum d]
            key = f"{name}/{dname}/"
            out[key + "x"] = npy(x).astype(np.uint8)
            out[key + "eps"] = npy(eps)
            out[key + "log_px"] = npy(log_px)
            out[key + "mi"] = npy(mi)
            out[key + "cov_norm"] = npy(cov_norm)
        # conv architecture (CIFAR shapes, soft targets in [0,1])
        B, n = 4, 4
        model, state0 = _build_reference_model("h2,s2,e2", "conv", 3072, 8192, B, False, False, 2.0, dtype)
        x = synthetic.uniform_batches(1, B, 3072, dtype=dtype)[0]
        ref_shim.reseed_eps(2100)
        with torch.no_grad():
            log_px, mi, cov_norm = model.log_likelihood(x, n=n)
        key = f"conv_h2s2e2/{dname}/"
        out[key + "x"] = npy(x)
        out[key + "eps"] = npy(torch.cat(list(ref_shim.eps_log), dim=-1))
        out[key + "log_px"], out[key + "mi"], out[key + "cov_norm"] = npy(log_px), npy(mi), npy(cov_norm)
    np.savez_compressed(os.path.join(HERE, "g4_loglik.npz"), **out)
    print("g4_loglik:", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- G5
def gen_parser():
    torch.set_default_dtype(torch.float32)
    table = {}
    for s in ["e6", "h2,s2,e2", "6h2,6s2,6e2", "e2,h2,s2", "h40", "s40-wn", "2h3,p2,d2,u2,c3", "3e2,2s5", "  H2 , S2 ",
              "p5", "e1", "d2,u2,p2", "3u2", "u2,d3,h2,s2,e2", "2d40"]:
        comps = ref_utils.parse_components(s, fixed_curvature=False)
        table[s] = {
            "canonical": ref_utils.canonical_name(comps),
            "components": [{
                "class": type(c).__name__, "dim": c.dim, "true_dim": c.true_dim, "shortcut": c._shortcut(),
                "params": [n for n, _ in c.named_parameters()],
            } for c in comps],
        }
    errors = {}
    for s in ["x2", "0h2", "h0", "h"]:
        try:
            ref_utils.parse_components(s, False)
            errors[s] = None
        except Exception as e:  # noqa: BLE001
            errors[s] = type(e).__name__
    betas = {
        "1,1,1,500": ref_utils.linear_betas(1.0, 1.0, 1, 500)[:4].tolist(),
        "0.1,2,5,8": ref_utils.linear_betas(0.1, 2.0, 5, 8).tolist(),
    }
    # parameter names/shapes of the headline models (state-dict contract)
    shapes = {}
    for s, arch, in_dim, h in [("h2,s2,e2", "ff", 784, 400), ("6h2,6s2,6e2", "ff", 784, 400), ("e6", "ff", 784, 400),
                               ("h2,s2,e2", "conv", 3072, 8192), ("u2,d2,e2", "ff", 784, 400)]:
        m, _ = _build_reference_model(s, arch, in_dim, h, 4, False, False, 2.0, torch.float32)
        shapes[f"{s}|{arch}"] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    with open(os.path.join(HERE, "g5_parser.json"), "w") as fh:
        json.dump({"parse": table, "errors": errors, "linear_betas": betas, "state_shapes": shapes}, fh, indent=1)
    print("g5_parser: ok")


if __name__ == "__main__":
    os.makedirs("/tmp/golden_chkpt", exist_ok=True)
    which = sys.argv[1:] or ["g1", "g1s", "g2", "g3s", "g3f", "g4", "g5", "g6", "g7", "g8"]
    torch.set_num_threads(8)
    if "g1" in which:
        gen_primitives()
    if "g1s" in which:
        gen_scalar_fns()
    if "g2" in which:
        gen_component()
    if "g3s" in which:
        gen_steps_small()
    if "g3f" in which:
        gen_steps_full()
    if "g4" in which:
        gen_loglik()
    if "g5" in which:
        gen_parser()
    if "g6" in which:
        gen_projected()
    if "g7" in which:
        gen_distances()
    if "g8" in which:
        gen_full_size_extra()
