"""Oracle for the projected sphere `d` and the universal manifold `u`.

Pinned part: every function of mt/mvae/ops/spherical_projected.py that does not cross into geoopt (g6_projected.npz,
recorded from the reference itself) and Universal.radius / _choice.  Unpinned part (mob_add -> geoopt mobius_add with
c = -K): checked through the properties the reference's own tests assert
(tests/mvae/ops/test_spherical_projected.py:86-271, test_universal.py), restated here on the same fixture points.
"""
import numpy as np
import pytest
import torch

from helpers import T, assert_close, load_npz
from mvae_amd import synthetic
from oracle import model as M
from oracle import ops as O

DT = {"f32": torch.float32, "f64": torch.float64}
RT = {"f32": 2e-5, "f64": 1e-9}
t64 = lambda *a: torch.tensor(a, dtype=torch.float64)  # noqa: E731
R2 = torch.tensor(2.0, dtype=torch.float64)
TEST_EPS = 5e-6  # test_spherical_projected.py:29


@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("d", [2, 5, 40])
@pytest.mark.parametrize("R", [0.5, 1.0, 2.0, 11.0])
def test_projected_sphere_pinned_functions(R, d, dname):
    g = load_npz("g6_projected.npz")
    k = f"D/R{R:g}/d{d}/{dname}/"
    x, y, w = T(g[k + "x"]), T(g[k + "y"]), T(g[k + "w"])
    Rt = torch.tensor(R, dtype=DT[dname])
    mu = O.d_exp_map_mu0(x, Rt)
    checks = [("mu", mu), ("log_mu0", O.d_log_map_mu0(mu, Rt)), ("pt", O.d_pt_mu0(x, y, Rt)),
              ("ipt", O.d_inv_pt_mu0(x, y, Rt)), ("lambda", O.d_lambda_x(y, Rt)),
              ("to_sphere", O.projected_to_spherical(y, Rt)),
              ("dist", O.spherical_projected_distance(y, w, 1 / Rt**2)), ("logdet", O.d_logdet(y, w, Rt)),
              ("logdet0", O.d_logdet(torch.zeros_like(w), w, Rt))]
    for name, val in checks:
        # the sphere's inverse map inside logdet divides two quantities that both vanish for nearby points
        assert_close(val.numpy(), g[k + name], RT[dname] * (20 if name.startswith("logdet") and dname == "f32" else 1),
                     k + name)


@pytest.mark.parametrize("dname", ["f32", "f64"])
def test_universal_radius_and_choice(dname):
    g = load_npz("g6_projected.npz")
    K = T(g[f"U/{dname}/K"])
    assert_close(torch.stack([O.u_radius(k) for k in K]).numpy(), g[f"U/{dname}/radius"], RT[dname], "u radius")
    assert [O.u_choice(k) for k in K] == list(g[f"U/{dname}/choice"])


# ---------------------------------------------------------------- properties (the reference's own test assertions)
def test_distance_definitions_agree():  # test_spherical_projected.py:113-120
    mu0, mu = t64(0., 0, 0), t64(2., 1., np.sqrt(2))
    K = 1 / R2**2
    gyr = O.spherical_projected_gyro_distance(mu0, mu, K)
    assert gyr.allclose(O.spherical_projected_distance(mu0, mu, K))
    back = O.projected_to_spherical  # distance on the sphere after back-projection (test_spherical.py helper)
    a, b = back(mu0, R2), back(mu, R2)
    sph = R2 * torch.acos(torch.clamp((a * b).sum() / R2**2, -1, 1))
    assert gyr.allclose(sph)
    for f in (O.spherical_projected_distance, O.spherical_projected_gyro_distance):
        assert f(mu, mu, K).allclose(t64(0.), atol=5e-4)
        assert f(mu0, mu, K) == f(mu, mu0, K)


@pytest.mark.parametrize("K", [1.0, 0.1, 10.0])
def test_mob_add_inverse(K):  # test_spherical_projected.py:123-129
    mu1 = t64(2., 1, np.sqrt(2))
    assert O.d_mob_add(mu1, -mu1, torch.tensor(K, dtype=torch.float64)).allclose(torch.zeros_like(mu1))


def test_parallel_transport_mu0_round_trip():  # test_spherical_projected.py:177-199
    mu0, mu2, u = t64(0., 0, 0), t64(np.sqrt(5), 1, np.sqrt(3)) / R2, t64(0, 2, -np.sqrt(2))
    assert O.d_pt_mu0(u, mu0, R2).allclose(u)
    assert O.d_inv_pt_mu0(O.d_pt_mu0(u, mu2, R2), mu2, R2).allclose(u)
    U = torch.stack((u, 2 * u))
    big = R2 * t64(np.sqrt(5), 1, np.sqrt(3))
    assert O.d_inv_pt_mu0(O.d_pt_mu0(U, big, R2), big, R2).allclose(U)


def test_exp_map_round_trips_and_geodesic_length():  # test_spherical_projected.py:202-252
    mu, u = t64(2., 1, np.sqrt(2)) / R2, t64(0, 2, -np.sqrt(2))
    z = O.d_exp_map(u, mu, R2)
    assert u.allclose(O.d_log_map(z, mu, R2), atol=TEST_EPS)
    c = 1 / R2**2
    assert O.spherical_projected_distance(mu, z, c).allclose(O.d_lambda_x_c(mu, c) * torch.norm(u, p=2))
    mu0 = torch.zeros(3, dtype=torch.float64)
    z0, z00 = O.d_exp_map(u, mu0, R2), O.d_exp_map_mu0(u, R2)
    assert z0.allclose(z00)
    assert u.allclose(O.d_log_map(z0, mu0, R2), atol=TEST_EPS) and u.allclose(O.d_log_map_mu0(z00, R2), atol=TEST_EPS)
    assert O.spherical_projected_distance(mu0, z0, c).allclose(2 * torch.norm(u, p=2))
    big_mu, big_u = t64(2., 1, np.sqrt(2)), 2.5 * t64(0, 2, -np.sqrt(2))  # test_exp_map_large
    assert big_u.allclose(O.d_log_map(O.d_exp_map(big_u, big_mu, R2), big_mu, R2), atol=TEST_EPS)
    U = torch.stack((u, 2 * u)) / R2  # test_exp_map_batch
    assert U.allclose(O.d_log_map(O.d_exp_map(U, mu, R2), mu, R2), atol=TEST_EPS)


def test_sample_projection_round_trip():  # test_spherical_projected.py:255-270
    v, mu = t64(0., 1, 2), t64(2., 1, np.sqrt(2)) / R2
    z, _ = O.d_sample_projection_mu0(v, mu, R2)
    _, v_ = O.d_inverse_sample_projection_mu0(z, mu, R2)
    assert v.allclose(v_, atol=TEST_EPS)


def test_projections_round_trip():  # test_spherical_projected.py:273-289
    assert O.projected_to_spherical(t64(0., 0), R2).allclose(R2 * t64(1., 0, 0))
    assert O.spherical_to_projected(R2 * t64(1., 0, 0), R2).allclose(t64(0., 0))
    mu_d = t64(1, np.sqrt(2)) / R2
    assert O.spherical_to_projected(O.projected_to_spherical(mu_d, R2), R2).allclose(mu_d)
    mu_s = t64(2., 1, np.sqrt(2))
    mu_s = mu_s / mu_s.norm() * R2
    assert O.projected_to_spherical(O.spherical_to_projected(mu_s, R2), R2).allclose(mu_s)


def test_exp_map_commutes_with_the_stereographic_projection():
    """Cross-model identity (the analogue of SURVEY 8c's lorentz_to_poincare check for `p`): walking a geodesic in the
    projected model and on the sphere ends at the same point."""
    g = torch.Generator().manual_seed(5)
    for R in (0.7, 2.0):
        Rt = torch.tensor(R, dtype=torch.float64)
        mu = torch.randn(16, 3, generator=g, dtype=torch.float64) * 0.4 * R
        u = torch.randn(16, 3, generator=g, dtype=torch.float64) * 0.3
        z = O.d_exp_map(u, mu, Rt)
        a, b = O.projected_to_spherical(mu, Rt), O.projected_to_spherical(z, Rt)
        sphere_dist = Rt * torch.acos(torch.clamp((a * b).sum(-1, keepdim=True) / Rt**2, -1, 1))
        assert sphere_dist.allclose(O.d_lambda_x(mu, Rt) * u.norm(dim=-1, keepdim=True), atol=2e-4)


# ---------------------------------------------------------------- universal component
@pytest.mark.parametrize("K,twin", [(-0.25, "p"), (0.25, "d"), (0.0, "e"), (5e-7, "e")])
def test_universal_component_equals_its_sub_manifold(K, twin):
    """universal.py:63-74 / sampling_procedures.py:184-206: `u` with curvature K is the Poincare ball (K < -eps), the
    projected sphere (K > eps) or the Euclidean normal procedure, at radius relu(1/sqrt|K|)."""
    g = torch.Generator().manual_seed(11)
    m, lv, eps = (torch.randn(6, 3, generator=g) * 0.5 for _ in range(3))
    Kt = torch.tensor(K)
    a = M.component_forward(M.ComponentSpec("u", 3), m, lv, eps, Kt)
    b = M.component_forward(M.ComponentSpec(twin, 3), m, lv, eps, None if twin == "e" else O.u_radius(Kt))
    assert torch.equal(a.z, b.z) and torch.equal(a.kl, b.kl)


def test_universal_curvature_gradient_is_clipped():
    """vae.py:161-163: the joint L2 norm of all `_curvature` gradients is clipped to 1 before the optimizer step; the
    SGD step on them follows CurvatureOptimizer's gate (train.py:357-358)."""
    spec = M.Spec("2u2,h2", in_dim=32, h_dim=16, fixed_curvature=False)
    st = synthetic.synthetic_state(spec.named_shapes(), radius=2.0)
    st["components.0._curvature"] = torch.tensor(-8.0)  # tiny radius -> large gradient
    st["components.1._curvature"] = torch.tensor(6.0)
    x, eps = synthetic.binary_batches(1, 8, 32)[0], synthetic.eps_batches(1, 8, 6)[0] * 3
    names = [f"components.{i}._curvature" for i in (0, 1)]
    free = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    raw = torch.stack(torch.autograd.grad(-M.forward(spec, free, x, eps).elbo, [free[n] for n in names]))
    assert float(raw.norm()) > 1.0  # otherwise the case does not exercise the clip
    orc = M.StepOracle(spec, st)
    orc.train_step(x, eps, 1.0, epoch=12)
    gk = torch.stack([orc.P[n].grad for n in names])
    assert torch.allclose(gk, raw / (raw.norm() + 1e-6), rtol=1e-6)  # clip_grad_norm_: g * max_norm / (norm + 1e-6)
    for n, g in zip(names, gk):
        assert torch.equal(orc.P[n].detach(), st[n] - 1e-4 * g)  # SGD(lr=1e-4): param.add_(grad, alpha=-lr)


# ---------------------------------------------------------------- Poincare ball: the same anchors
def test_poincare_round_trips_at_the_reference_tolerances():
    """tests/mvae/ops/test_poincare.py:140-211 restated: these tolerances are what anchors the mobius_add denominator
    guard (clamp_min(1e-15)); the '+ 1e-5' variant misses them by 1e-4 ... 0.45."""
    R = torch.tensor(2., dtype=torch.float32)
    # as in the reference, np.sqrt makes these float64 tensors (torch.tensor of mixed Python / numpy floats)
    mu, u = torch.tensor([2., 1, np.sqrt(2)]) / R, torch.tensor([0, 2, -np.sqrt(2)])
    assert mu.dtype == torch.float64
    z = O.p_exp_map(u, mu, R)
    assert u.allclose(O.p_log_map(z, mu, R), atol=TEST_EPS)  # test_exp_map
    c = 1 / R**2
    sub = O.p_mobius_add(-mu, z, c).norm()
    dist = 2 / c.sqrt() * torch.atanh(c.sqrt() * sub)  # poincare_distance (poincare.py:92-105 -> geoopt dist)
    assert dist.allclose(O.p_lambda_x(mu, c) * torch.norm(u, p=2), rtol=1e-4)
    mu0 = torch.zeros(3, dtype=torch.float64)
    z0 = O.p_exp_map(u, mu0, R)
    assert z0.allclose(O.p_exp_map_mu0(u, R))  # test_exp_map_mu0
    assert u.allclose(O.p_log_map(z0, mu0, R), atol=TEST_EPS)
    mud, ud = mu.double(), 2.5 * u.double()  # test_exp_map_large
    assert ud.allclose(O.p_log_map(O.p_exp_map(ud, mud, R.double()), mud, R.double()), atol=1e-4)
    U = torch.stack((u.double(), 2 * u.double()))  # test_exp_map_batch
    assert U.allclose(O.p_log_map(O.p_exp_map(U, mud, R.double()), mud, R.double()), atol=1e-5)
    v = torch.tensor([0., 1, 2], dtype=torch.float64)  # test_sample_projection
    zz, _ = O.p_sample_projection_mu0(v, mu, R)
    assert float(zz.dot(zz)) <= (1 + 1e-4) * 4.0
    _, v_ = O.p_inverse_sample_projection_mu0(zz, mu, R)
    assert v.allclose(v_, atol=TEST_EPS)
