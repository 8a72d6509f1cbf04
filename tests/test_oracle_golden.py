"""Pins the oracle (CPU restatement) to vectors recorded from the reference itself (tests/golden/make_golden.py).

CPU-only.  Bars: f64 1e-9 relative (same formulas, same library => differences are op-ordering only);
f32 2e-5 relative at primitive level, 1e-4 at step level (the north-star bar).
"""
import numpy as np
import pytest
import torch

from helpers import T, assert_close, load_json, load_npz, summary_of
from mvae_amd import synthetic
from oracle import model as M
from oracle import ops as O

DT = {"f32": torch.float32, "f64": torch.float64}
RT = {"f32": 2e-5, "f64": 1e-9}


# ------------------------------------------------------------------------------------------------ G1 scalar functions
@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("name", ["acosh", "atanh", "cosh", "sinh", "sqrt", "logsinh", "logcosh", "clamp_m1_2"])
def test_scalar_functions_and_custom_gradients(name, dname):
    g = load_npz("g1_scalar_fns.npz")
    fn = {
        "acosh": O.acosh, "atanh": O.atanh, "cosh": O.cosh, "sinh": O.sinh, "sqrt": O.sqrt, "logsinh": O.logsinh,
        "logcosh": O.logcosh, "clamp_m1_2": lambda t: O.leaky_clamp(t, -1.0, 2.0),
    }[name]
    x = T(g[f"{name}/{dname}/x"]).requires_grad_(True)
    y = fn(x)
    (dx,) = torch.autograd.grad(y.sum(), x)
    assert_close(y.detach().numpy(), g[f"{name}/{dname}/y"], RT[dname], f"{name} value")
    assert_close(dx.numpy(), g[f"{name}/{dname}/dy"], RT[dname], f"{name} grad")


def test_acosh_known_answers():
    """Reference tests/mvae/ops/test_common.py:45-50: acosh == np.arccosh on 100 seeded values (float32)."""
    g = load_npz("g1_scalar_fns.npz")
    got = O.acosh(torch.tensor(g["acosh_known/x"]).float()).numpy()
    assert np.allclose(np.arccosh(g["acosh_known/x"]), got)
    assert_close(got, g["acosh_known/y_f32"], 1e-6, "acosh vs reference f32")


@pytest.mark.parametrize("fn", [O.acosh, O.sqrt, O.cosh, O.sinh, O.logsinh, O.logcosh])
def test_scalar_functions_finite(fn):
    """Reference test_common.py:23-42: finite on +-500."""
    np.random.seed(42)
    xs = (np.random.random_sample(100) - 0.5) * 1000.0
    assert torch.isfinite(fn(torch.tensor(xs).float())).all()


# ------------------------------------------------------------------------------------------------ G1 primitives
@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("d", [2, 5, 40])
@pytest.mark.parametrize("R", [0.5, 1.0, 2.0, 11.0])
@pytest.mark.parametrize("man", ["H", "S", "E"])
def test_primitives(man, R, d, dname):
    g = load_npz("g1_primitives.npz")
    k = f"{man}/R{R:g}/d{d}/{dname}/"
    x, v = T(g[k + "x"]), T(g[k + "v"])
    rt = RT[dname]
    if man == "E":
        mu = O.e_exp_map_mu0(x)
        z, (u, _) = O.e_sample_projection_mu0(v, mu)
        iu, iv = O.e_inverse_sample_projection_mu0(z, mu)
        for name, val in [("mu", mu), ("z", z), ("u", u), ("inv_u", iu), ("inv_v", iv),
                          ("log_mu0", O.e_log_map_mu0(mu))]:
            assert_close(val.numpy(), g[k + name], rt, k + name)
        return
    Rt = torch.tensor(R, dtype=DT[dname])
    f = {n: getattr(O, ("h_" if man == "H" else "s_") + n) for n in
         ["exp_map_mu0", "pt_mu0", "inv_pt_mu0", "exp_map", "log_map", "log_map_mu0", "sample_projection_mu0",
          "inverse_sample_projection_mu0", "logdet"]}
    mu = f["exp_map_mu0"](x, Rt)
    z, (u, _) = f["sample_projection_mu0"](v, mu, Rt)
    iu, iv = f["inverse_sample_projection_mu0"](z, mu, Rt)
    mu0 = O.h_mu0(mu.shape, Rt)
    u0, v0 = f["inverse_sample_projection_mu0"](z, mu0, Rt)
    checks = [("mu", mu), ("z", z), ("u", u), ("inv_u", iu), ("inv_v", iv), ("inv0_u", u0), ("inv0_v", v0),
              ("logdet_u", f["logdet"](u, Rt)), ("logdet_u0", f["logdet"](u0, Rt)),
              ("pt", f["pt_mu0"](O._prepend_zero(v), mu, Rt)), ("ipt", f["inv_pt_mu0"](u, mu, Rt)),
              ("exp", f["exp_map"](u, mu, Rt)), ("log", f["log_map"](z, mu, Rt)),
              ("log_mu0", f["log_map_mu0"](mu, Rt))]
    if man == "H":
        checks.append(("to_poincare", O.lorentz_to_poincare(mu, Rt)))
    else:
        checks.append(("to_projected", O.spherical_to_projected(mu, Rt)))
    for name, val in checks:
        # f32: inverse maps of points far out on the hyperboloid are ill-conditioned (cancellation in <.,.>_L),
        # both sides evaluate the same expression so they still agree tightly
        assert_close(val.numpy(), g[k + name], rt, k + name)


def test_reference_known_answers():
    """Known answers restated (values, not code) from reference tests/mvae/ops/test_hyperbolics.py:59-71,80-83,104."""
    t = lambda *a: torch.tensor(a, dtype=torch.float32)  # noqa: E731
    assert O.lorentz_product(t(0, 0, 0), t(3, 2, 1)) == 0
    assert O.lorentz_product(t(1, 2, 3), t(3, 2, 1)) == 4
    assert O.lorentz_product(t(1, 2, 3), t(0, 2, 1)) == 7
    assert O.lorentz_product(t(1, 2, 3), t(0, 0, 0)) == 0
    assert torch.equal(O.h_mu0((3, 3), torch.tensor(2.0)), 2.0 * torch.tensor([[1.0, 0, 0]] * 3))
    assert O.lorentz_norm(t(2, 1, 2)).allclose(torch.tensor(1.0))
    assert torch.isfinite(O.lorentz_norm(t(300, 0, 0)))
    # R=2 fixture of the reference op tests: mu = R(2,1,sqrt2), u = (0,2,-sqrt2): exp o log = id, dist = |u|
    R = torch.tensor(2.0)
    mu = R * t(2, 1, 2**0.5)
    u = t(0, 2, -2**0.5)
    zz = O.h_exp_map(u, mu, R)
    assert u.allclose(O.h_log_map(zz, mu, R), atol=5e-6)
    dist = R * O.acosh(-O.lorentz_product(mu, zz) / R**2)
    assert dist.allclose(torch.norm(u))
    v = t(1, 2)
    zp, (pt, _) = O.h_sample_projection_mu0(v, mu, R)
    assert abs(float(O.lorentz_product(pt, mu))) < 5e-6  # tangent at mu
    assert abs(float(O.lorentz_product(zp, zp) + R * R)) < 1e-4  # on the hyperboloid
    _, v_ = O.h_inverse_sample_projection_mu0(zp, mu, R)
    assert v.allclose(v_, atol=5e-6)


# ------------------------------------------------------------------------------------------------ G2 component
def _g2_keys():
    g = load_npz("g2_component.npz")
    return sorted({k.rsplit("/", 1)[0] for k in g})


@pytest.mark.parametrize("key", _g2_keys())
def test_component_forward_and_gradients(key):
    g = load_npz("g2_component.npz")
    comp, Rs, _, dname = key.split("/")
    c = M.ComponentSpec(comp[0], int(comp[1:]))
    dt = DT[dname]
    mean_raw = T(g[key + "/mean_raw"]).requires_grad_(True)
    lv_raw = T(g[key + "/logvar_raw"]).requires_grad_(True)
    eps = T(g[key + "/eps"])
    rp = torch.tensor(float(Rs[1:]), dtype=dt, requires_grad=True) if c.radius_name else None
    o = M.component_forward(c, mean_raw, lv_raw, eps, rp)
    loss = (T(g[key + "/wz"]) * o.z).sum() + (T(g[key + "/wkl"]) * o.kl).sum()
    params = [mean_raw, lv_raw] + ([rp] if rp is not None else [])
    grads = torch.autograd.grad(loss, params)
    rt = RT[dname] * 5
    assert_close(o.mu.detach().numpy(), g[key + "/mu"], rt, "mu")
    assert_close(o.z.detach().numpy(), g[key + "/z"], rt, "z")
    assert_close(o.kl.detach().numpy(), g[key + "/kl"], rt, "kl")
    if c.letter != "e":
        assert_close(o.u.detach().numpy(), g[key + "/u"], rt, "u")
        assert_close(o.log_q.detach().numpy(), g[key + "/logq"], rt, "logq")
        assert_close(o.log_p.detach().numpy(), g[key + "/logp"], rt, "logp")
    assert_close(grads[0].numpy(), g[key + "/d_mean_raw"], rt, "d_mean_raw")
    assert_close(grads[1].numpy(), g[key + "/d_logvar_raw"], rt, "d_logvar_raw")
    if rp is not None:
        assert_close(grads[2].numpy(), g[key + "/d_radius"], rt, "d_radius")


# ------------------------------------------------------------------------------------------------ G3 steps (small)
SMALL = load_json("g3_step_small.json")


def _small_case(name, dname, steps):
    g = load_npz("g3_step_small.npz")
    meta = SMALL[name]
    key = f"{name}/{dname}/steps{steps}/"
    if key + "stats" not in g:
        pytest.skip("case not stored for this dtype")
    spec = M.Spec(meta["model"], in_dim=meta["in_dim"], h_dim=meta["h_dim"],
                  scalar_parametrization=meta["scalar_parametrization"], fixed_curvature=meta["fixed_curvature"])
    dt = DT[dname]
    state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0, dtype=dt)
    return g, meta, key, spec, dt, state0


@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("name", sorted(SMALL))
def test_train_step_small_one_step(name, dname):
    g, meta, key, spec, dt, state0 = _small_case(name, dname, 1)
    for k, v in state0.items():  # the recipe state IS the state the reference ran from
        np.testing.assert_array_equal(v.numpy(), g[key + "state0/" + k])
    orc = M.StepOracle(spec, state0, dtype=dt)
    orc.begin_epoch(meta["epoch"])
    x = T(g[key + "x"], dt)[0]
    eps = T(g[key + "eps"], dt)[0]
    out = orc.train_step(x, eps, beta=1.0, epoch=meta["epoch"])
    rt = 1e-4 if dname == "f32" else 1e-9
    assert_close(out.logits.detach().numpy(), g[key + "logits"], rt, "logits")
    assert_close(out.concat_z.detach().numpy(), g[key + "concat_z"], rt, "concat_z")
    assert_close(out.bce.detach().numpy(), g[key + "bce_rows"], rt, "bce")
    assert_close(out.kl.detach().numpy(), g[key + "kl_rows"], rt, "kl")
    st = g[key + "stats"][0]
    assert_close(float(out.elbo), st[2], rt, "elbo")
    assert_close(out.kl.sum(dim=1).detach().numpy(), st[3:], rt, "component_kl sums")
    for n, p in orc.P.items():
        if key + "grad/" + n in g:
            assert_close(p.grad.numpy(), g[key + "grad/" + n], rt, "grad " + n)
        else:
            assert p.grad is None
        assert_close(p.detach().numpy(), g[key + "state1/" + n], rt, "state1 " + n)


@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("name", sorted(SMALL))
def test_train_step_small_five_steps(name, dname):
    g, meta, key, spec, dt, state0 = _small_case(name, dname, 5)
    orc = M.StepOracle(spec, state0, dtype=dt)
    orc.begin_epoch(meta["epoch"])
    xs, eps = T(g[key + "x"], dt), T(g[key + "eps"], dt)
    rt = 2e-4 if dname == "f32" else 1e-8
    for s in range(5):
        out = orc.train_step(xs[s], eps[s], beta=1.0, epoch=meta["epoch"])
        st = g[key + "stats"][s]
        for got, want, nm in zip([out.bce.sum(), out.kl.sum(), out.elbo], st[:3], ["bce", "kl", "elbo"]):
            assert_close(float(got), want, rt, f"{nm} step {s}")
    for n, p in orc.P.items():
        assert_close(p.detach().numpy(), g[key + "state_final/" + n], rt, "state_final " + n)


# ------------------------------------------------------------------------------------------------ G3 steps (full size)
FULL = load_json("g3_step_full.json")


@pytest.mark.parametrize("name", sorted(FULL))
def test_train_step_full_size_f32(name):
    g = load_npz("g3_step_full.npz")
    meta = FULL[name]
    spec = M.Spec(meta["model"], in_dim=meta["in_dim"], h_dim=meta["h_dim"], arch=meta["arch"],
                  fixed_curvature=meta["fixed_curvature"])
    dt = torch.float32
    tconv = ("d1", "d2", "d3") if meta["arch"] == "conv" else ()
    state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0, dtype=dt, transposed_conv=tconv)
    gen = synthetic.uniform_batches if meta["soft_targets"] else synthetic.binary_batches
    rt = 1e-4
    for steps in (1, 5):
        key = f"{name}/f32/steps{steps}/"
        xs = gen(steps, meta["batch"], meta["in_dim"], dtype=dt)
        eps = synthetic.eps_batches(steps, meta["batch"], spec.total_true_dim, dtype=dt)
        orc = M.StepOracle(spec, state0, dtype=dt)
        orc.begin_epoch(meta["epoch"])
        for s in range(steps):
            out = orc.train_step(xs[s], eps[s], beta=1.0, epoch=meta["epoch"])
            st = g[key + "stats"][s]
            for got, want, nm in zip([out.bce.sum(), out.kl.sum(), out.elbo], st[:3], ["bce", "kl", "elbo"]):
                assert_close(float(got), want, rt, f"{nm} step {s}")
            if steps == 1:
                assert_close(out.concat_z.detach().numpy(), g[key + "concat_z"], rt, "concat_z")
                assert_close(out.bce.detach().numpy(), g[key + "bce_rows"], rt, "bce rows")
                assert_close(out.kl.detach().numpy(), g[key + "kl_rows"], rt, "kl rows")
                ref = g[key + "logits_summary"]
                assert_close(summary_of(out.logits.detach().numpy(), ref), ref, rt, "logits summary")
                for n, p in orc.P.items():
                    if key + "grad_summary/" + n in g:
                        ref = g[key + "grad_summary/" + n]
                        assert_close(summary_of(p.grad.numpy(), ref), ref, rt, "grad " + n, atol_frac=1e-4)
        for n, p in orc.P.items():
            ref = g[key + "state_final_summary/" + n]
            assert_close(summary_of(p.detach().numpy(), ref), ref, rt, "final " + n, atol_frac=1e-4)


# ------------------------------------------------------------------------------------------------ G8 (full size, extra)
EXTRA = load_json("g8_full_size_extra.json")


@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("name", sorted(EXTRA))
def test_train_step_full_size_extra(name, dname):
    """The oracle against one reference step of BASELINE config [4] at its real batch (B = 256) and of the reference's
    large-component models `h40` / `s40` at the benchmark size: float64 to 1e-9, float32 to 1e-4."""
    g = load_npz("g8_full_size_extra.npz")
    meta = EXTRA[name]
    spec = M.Spec(meta["model"], in_dim=meta["in_dim"], h_dim=meta["h_dim"], arch=meta["arch"],
                  fixed_curvature=meta["fixed_curvature"])
    dt = torch.float32 if dname == "f32" else torch.float64
    rt = 1e-4 if dname == "f32" else 1e-9
    tconv = ("d1", "d2", "d3") if meta["arch"] == "conv" else ()
    state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0, dtype=dt, transposed_conv=tconv)
    gen = synthetic.uniform_batches if meta["soft_targets"] else synthetic.binary_batches
    key = f"{name}/{dname}/"
    x = gen(1, meta["batch"], meta["in_dim"], dtype=dt)[0]
    eps = synthetic.eps_batches(1, meta["batch"], spec.total_true_dim, dtype=dt)[0]
    torch.set_num_threads(min(8, torch.get_num_threads()))
    orc = M.StepOracle(spec, state0, dtype=dt)
    orc.begin_epoch(meta["epoch"])
    out = orc.train_step(x, eps, beta=1.0, epoch=meta["epoch"])
    st = g[key + "stats"][0]
    for got, want, nm in zip([out.bce.sum(), out.kl.sum(), out.elbo], st[:3], ["bce", "kl", "elbo"]):
        assert_close(float(got), want, rt, nm)
    assert_close(out.concat_z.detach().numpy(), g[key + "concat_z"], rt, "concat_z")
    assert_close(out.bce.detach().numpy(), g[key + "bce_rows"], rt, "bce rows")
    assert_close(out.kl.detach().numpy(), g[key + "kl_rows"], rt, "kl rows", atol_frac=rt)
    ref = g[key + "logits_summary"]
    assert_close(summary_of(out.logits.detach().numpy(), ref), ref, rt, "logits summary")
    for n, p in orc.P.items():
        if key + "grad_summary/" + n in g:
            ref = g[key + "grad_summary/" + n]
            assert_close(summary_of(p.grad.numpy(), ref), ref, rt, "grad " + n, atol_frac=rt)
        ref = g[key + "state1_summary/" + n]
        assert_close(summary_of(p.detach().numpy(), ref), ref, max(rt, 1e-7), "param " + n, atol_frac=max(rt, 1e-7))


# ------------------------------------------------------------------------------------------------ G4 log-likelihood
@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("name,model", [("h2s2e2", "h2,s2,e2"), ("e6", "e6"), ("h5s3e4", "h5,s3,e4")])
def test_log_likelihood(name, model, dname):
    g = load_npz("g4_loglik.npz")
    dt = DT[dname]
    spec = M.Spec(model, in_dim=32, h_dim=16, fixed_curvature=False)
    P = synthetic.synthetic_state(spec.named_shapes(), radius=2.0, dtype=dt)
    key = f"{name}/{dname}/"
    with torch.no_grad():
        lp, mi, cn = M.log_likelihood(spec, P, T(g[key + "x"], dt), T(g[key + "eps"], dt))
    rt = 1e-4 if dname == "f32" else 1e-9
    assert_close(lp.numpy(), g[key + "log_px"], rt, "log_px")
    assert_close(mi.numpy(), g[key + "mi"], rt, "mi")
    assert_close(float(cn), float(g[key + "cov_norm"]), rt, "cov_norm")


@pytest.mark.parametrize("dname", ["f32", "f64"])
def test_log_likelihood_conv(dname):
    g = load_npz("g4_loglik.npz")
    dt = DT[dname]
    spec = M.Spec("h2,s2,e2", in_dim=3072, h_dim=8192, arch="conv", fixed_curvature=False)
    P = synthetic.synthetic_state(spec.named_shapes(), radius=2.0, dtype=dt, transposed_conv=("d1", "d2", "d3"))
    key = f"conv_h2s2e2/{dname}/"
    with torch.no_grad():
        lp, mi, cn = M.log_likelihood(spec, P, T(g[key + "x"], dt), T(g[key + "eps"], dt))
    rt = 2e-4 if dname == "f32" else 1e-9
    assert_close(lp.numpy(), g[key + "log_px"], rt, "log_px")
    assert_close(mi.numpy(), g[key + "mi"], rt, "mi", atol_frac=1e-3 if dname == "f32" else None)
    assert_close(float(cn), float(g[key + "cov_norm"]), rt, "cov_norm")


# ------------------------------------------------------------------------------------------------ G5 parser
def test_parser_table():
    tab = load_json("g5_parser.json")
    for s, ref in tab["parse"].items():
        if any(c["class"] == "ConstantComponent" for c in ref["components"]):  # `c` is not built
            with pytest.raises(NotImplementedError):
                M.parse_components(s)
            continue
        comps = M.parse_components(s)
        assert [(c.letter, c.dim, c.true_dim) for c in comps] == \
            [(r["shortcut"][0], r["dim"], r["true_dim"]) for r in ref["components"]]
    for s, err in tab["errors"].items():
        with pytest.raises({"ValueError": ValueError, "NotImplementedError": NotImplementedError}[err]):
            M.parse_components(s)
    for key, shapes in tab["state_shapes"].items():
        model, arch = key.split("|")
        spec = M.Spec(model, in_dim=3072 if arch == "conv" else 784, h_dim=8192 if arch == "conv" else 400, arch=arch)
        assert [[n, list(s)] for n, s in spec.named_shapes()] == shapes


# ----------------------------------------------------------------------------------------------- G7 distances
@pytest.mark.parametrize("dname,tol", [("f64", 1e-9), ("f32", 2e-5)])
def test_distances_vs_reference_helpers(dname, tol):
    """oracle h/s/e geodesic distances, Lorentz product and norm == the reference's own helpers
    (tests/mvae/ops/test_{hyperbolics,spherical,euclidean}.py, recorded by make_golden.gen_distances)."""
    from oracle import ops
    g = load_npz("g7_distances.npz")
    dt = torch.float64 if dname == "f64" else torch.float32
    n = 0
    for mname in "HSE":
        for R in ["0.5", "1", "2", "11"]:
            for d in [2, 5, 40]:
                k = f"{mname}/R{R}/d{d}/{dname}/"
                p, q, Rt = T(g[k + "p"], dt), T(g[k + "q"], dt), torch.tensor(float(R), dtype=dt)
                if mname == "H":
                    assert_close(ops.h_exp_map_mu0(T(g[k + "x"], dt), Rt).numpy(), g[k + "p"], tol, k + "p")
                    assert_close(ops.h_distance(p, q, Rt).numpy(), g[k + "dist"], tol, k + "dist", atol_frac=tol)
                    assert_close(ops.lorentz_product(p, q, keepdim=True).numpy(), g[k + "lprod"], tol, k + "lprod")
                    u = ops.h_log_map(q, p, Rt)
                    assert_close(ops.lorentz_norm(u, keepdim=True).numpy(), g[k + "lnorm_u"], 50 * tol, k + "lnorm_u")
                elif mname == "S":
                    assert_close(ops.s_distance(p, q, Rt).numpy(), g[k + "dist"], tol, k + "dist", atol_frac=tol)
                else:
                    assert_close(ops.e_distance(p, q).numpy(), g[k + "dist"], tol, k + "dist")
                n += 1
    assert n == 36
