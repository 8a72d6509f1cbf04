"""Pins the parity-UNPINNED arithmetic (Poincare ball `p`; `mob_add` inside the projected sphere `d`) THROUGH the pinned
models, on the CPU oracle in float64 to 1e-9 -- values and gradients.

geoopt 0.1.0 (where `pm.expmap`, `pm.logmap`, `pm.mobius_add`, ... live) cannot be obtained, so no reference output
exists for those functions.  But the ball is isometric to the hyperboloid through the reference-owned
`poincare_to_lorentz` (rho, poincare.py:167-170), and the projected sphere to the sphere through
`projected_to_spherical` (sigma, spherical_projected.py:191-196); the hyperboloid and the sphere ARE pinned to vectors
recorded from the reference (g1/g2).  An isometry maps geodesics to geodesics and commutes with exp / log / parallel
transport, so every p / d operator is determined by its h / s counterpart:

    rho(exp0_p(x))            = exp0_h(2 x)                      (lambda_0 = 2: |x| in the chart is 2|x| of arc length)
    rho(exp_p(u, at))         = exp_h(d rho_at(u), rho(at))      d rho_y(u) = the differential of rho at y, closed form
    d rho_at(log_p(z, at))    = log_h(rho(z), rho(at))                        below (checked against autograd's JVP)
    rho(sample_p(v, at)[0])   = sample_h(v, rho(at))[0]          (SAME v: PT_0->at (v/2) in the ball <-> PT([0, v]))
    dist_p(x, y)              = dist_h(rho(x), rho(y))
    z, KL of a `p` component with heads (m, l) = rho^-1 / KL of an `h` component with heads (2m, l), same eps and R,
    hence d KL/dm|_p = 2 d KL/dm|_h(2m), d KL/dl and d KL/dR equal          -- and the same with sigma, s for `d`.

A wrong guard constant, sign or factor in the restated gyrovector formulas breaks these identities at O(1); they hold
to rounding (1e-9 relative in float64 away from the guards).  tests/test_ops_gpu.py runs the same identities on the HIP
kernels at 1e-4 (float32).
"""
import numpy as np
import pytest
import torch

from helpers import assert_close
from oracle import model as M
from oracle import ops

F64 = torch.float64
TOL = 1e-9


def _pts(seed, rows, d, R, scale=0.35):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(rows, d, generator=g, dtype=F64) * scale * min(R, 3.0) / np.sqrt(d))


def d_rho(y, u, R):
    """Differential of poincare_to_lorentz at y applied to u (closed form).  rho(y) = [R(R^2+n), 2R^2 y] / (R^2-n),
    n = |y|^2."""
    n = (y * y).sum(-1, keepdim=True)
    yu = (y * u).sum(-1, keepdim=True)
    D = R * R - n
    d0 = 4 * R**3 * yu / D**2
    di = 2 * R * R * u / D + 4 * R * R * y * yu / D**2
    return torch.cat((d0, di), dim=-1)


def d_sigma(y, u, R):
    """Differential of projected_to_spherical at y applied to u.  sigma(y) = [R(R^2-n), 2R^2 y] / (R^2+n)."""
    n = (y * y).sum(-1, keepdim=True)
    yu = (y * u).sum(-1, keepdim=True)
    D = R * R + n
    d0 = -4 * R**3 * yu / D**2
    di = 2 * R * R * u / D - 4 * R * R * y * yu / D**2
    return torch.cat((d0, di), dim=-1)


CASES = [(R, d) for R in (0.5, 1.0, 2.0, 11.0) for d in (2, 5)]


@pytest.mark.parametrize("R,d", CASES)
def test_differentials_match_autograd_jvp(R, d):
    Rt = torch.tensor(R, dtype=F64)
    y, u = ops.p_exp_map_mu0(_pts(1, 6, d, R), Rt), _pts(2, 6, d, R)
    for conv, diff in ((ops.poincare_to_lorentz, d_rho), (ops.projected_to_spherical, d_sigma)):
        _, jvp = torch.autograd.functional.jvp(lambda t: conv(t, Rt), (y,), (u,))
        assert_close(diff(y, u, Rt).numpy(), jvp.numpy(), TOL, "differential")


@pytest.mark.parametrize("R,d", CASES)
def test_poincare_maps_are_the_hyperboloid_maps(R, d):
    Rt = torch.tensor(R, dtype=F64)
    x, u, v = _pts(3, 8, d, R), _pts(4, 8, d, R), _pts(5, 8, d, R)
    at = ops.p_exp_map_mu0(x, Rt)
    rho = lambda t: ops.poincare_to_lorentz(t, Rt)  # noqa: E731
    assert_close(rho(at).numpy(), ops.h_exp_map_mu0(2 * x, Rt).numpy(), TOL, "exp_map_mu0")
    assert_close((2 * ops.p_log_map_mu0(at, Rt)).numpy(), ops.h_log_map_mu0(rho(at), Rt)[..., 1:].numpy(), TOL,
                 "inverse_exp_map_mu0", atol_frac=TOL)
    z = ops.p_exp_map(u, at, Rt)
    assert_close(rho(z).numpy(), ops.h_exp_map(d_rho(at, u, Rt), rho(at), Rt).numpy(), TOL, "exp_map")
    assert_close(d_rho(at, ops.p_log_map(z, at, Rt), Rt).numpy(), ops.h_log_map(rho(z), rho(at), Rt).numpy(), 1e-8,
                 "inverse_exp_map", atol_frac=1e-8)
    zs, (us, _) = ops.p_sample_projection_mu0(v, at, Rt)
    zh, (uh, _) = ops.h_sample_projection_mu0(v, rho(at), Rt)
    assert_close(rho(zs).numpy(), zh.numpy(), TOL, "sample_projection z")
    assert_close(d_rho(at, us, Rt).numpy(), uh.numpy(), TOL, "sample_projection u", atol_frac=TOL)
    _, v_back = ops.p_inverse_sample_projection_mu0(zs, at, Rt)
    assert_close(v_back.numpy(), v.numpy(), 1e-8, "inverse_sample_projection", atol_frac=1e-8)
    # parallel transport from the origin: PT_0->at(w) in the ball <-> PT_mu0->rho(at)([0, 2w])
    w = _pts(6, 8, d, R)
    pt_h = ops.h_pt_mu0(torch.cat((torch.zeros(8, 1, dtype=F64), 2 * w), -1), rho(at), Rt)
    assert_close(d_rho(at, ops.p_pt_mu0(w, at, Rt), Rt).numpy(), pt_h.numpy(), TOL, "parallel_transport_mu0",
                 atol_frac=TOL)
    y2 = ops.p_exp_map_mu0(_pts(7, 8, d, R), Rt)
    assert_close(ops.p_distance(at, y2, Rt).numpy(), ops.h_distance(rho(at), rho(y2), Rt).numpy(), 1e-8, "distance",
                 atol_frac=1e-8)


@pytest.mark.parametrize("R,d", CASES)
def test_projected_sphere_maps_are_the_sphere_maps(R, d):
    Rt = torch.tensor(R, dtype=F64)
    x, u, v = _pts(13, 8, d, R, 0.25), _pts(14, 8, d, R, 0.25), _pts(15, 8, d, R, 0.25)
    at = ops.d_exp_map_mu0(x, Rt)
    sig = lambda t: ops.projected_to_spherical(t, Rt)  # noqa: E731
    assert_close(sig(at).numpy(), ops.s_exp_map_mu0(2 * x, Rt).numpy(), TOL, "exp_map_mu0")
    z = ops.d_exp_map(u, at, Rt)
    assert_close(sig(z).numpy(), ops.s_exp_map(d_sigma(at, u, Rt), sig(at), Rt).numpy(), TOL, "exp_map")
    assert_close(d_sigma(at, ops.d_log_map(z, at, Rt), Rt).numpy(), ops.s_log_map(sig(z), sig(at), Rt).numpy(), 1e-8,
                 "inverse_exp_map", atol_frac=1e-8)
    zs, _ = ops.d_sample_projection_mu0(v, at, Rt)
    zh, _ = ops.s_sample_projection_mu0(v, sig(at), Rt)
    assert_close(sig(zs).numpy(), zh.numpy(), TOL, "sample_projection z")
    y2 = ops.d_exp_map_mu0(_pts(17, 8, d, R, 0.25), Rt)
    K = 1 / Rt**2
    assert_close(ops.spherical_projected_gyro_distance(at, y2, K).numpy(), ops.s_distance(sig(at), sig(y2), Rt).numpy(),
                 1e-8, "gyro distance", atol_frac=1e-8)
    assert_close(ops.spherical_projected_distance(at, y2, K).numpy(), ops.s_distance(sig(at), sig(y2), Rt).numpy(), 1e-7,
                 "distance", atol_frac=1e-7)


def _component(letter, d, m, lv, eps, R):
    m, lv = m.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    Rp = torch.tensor(R, dtype=F64, requires_grad=True)
    out = M.component_forward(M.ComponentSpec(letter, d), m, lv, eps, Rp)
    return out, m, lv, Rp


@pytest.mark.parametrize("ball,ambient", [("p", "h"), ("d", "s")])
@pytest.mark.parametrize("R,d", CASES)
def test_component_value_and_gradients_through_the_pinned_model(ball, ambient, R, d):
    """z, KL and their gradients w.r.t. the mean head, the logvar head and the radius parameter of a `p` (`d`)
    component == those of the PINNED `h` (`s`) component fed 2 * mean (tangent-scale mapping lambda_0 = 2)."""
    g = torch.Generator().manual_seed(100 + d)
    m = torch.randn(8, d, generator=g, dtype=F64) * 0.3 * min(R, 2.0) / np.sqrt(d)
    lv = torch.randn(8, d, generator=g, dtype=F64) * 0.5 - 1.0
    eps = torch.randn(8, d, generator=g, dtype=F64)
    ob, mb, lb, Rb = _component(ball, d, m, lv, eps, R)
    oa, ma, la, Ra = _component(ambient, d, 2 * m, lv, eps, R)
    conv = ops.poincare_to_lorentz if ball == "p" else ops.projected_to_spherical
    assert_close(conv(ob.z, Rb).detach().numpy(), oa.z.detach().numpy(), TOL, "z")
    assert_close(ob.kl.detach().numpy(), oa.kl.detach().numpy(), 1e-8, "kl", atol_frac=1e-8)
    w = torch.randn(8, generator=g, dtype=F64)  # a generic upstream gradient on kl
    (ob.kl * w).sum().backward()
    (oa.kl * w).sum().backward()
    assert_close(mb.grad.numpy(), 2 * ma.grad.numpy(), 1e-7, "d kl / d mean", atol_frac=1e-7)
    assert_close(lb.grad.numpy(), la.grad.numpy(), 1e-7, "d kl / d logvar", atol_frac=1e-7)
    assert_close(Rb.grad.numpy(), Ra.grad.numpy(), 1e-6, "d kl / d radius", atol_frac=1e-6)
    # gradient of z: contract conv(z) with a fixed cotangent on both sides
    ob, mb, lb, Rb = _component(ball, d, m, lv, eps, R)
    oa, ma, la, Ra = _component(ambient, d, 2 * m, lv, eps, R)
    ct = torch.randn(8, d + 1, generator=g, dtype=F64)
    (conv(ob.z, Rb) * ct).sum().backward()
    (oa.z * ct).sum().backward()
    assert_close(mb.grad.numpy(), 2 * ma.grad.numpy(), 1e-7, "d z / d mean", atol_frac=1e-7)
    assert_close(lb.grad.numpy(), la.grad.numpy(), 1e-7, "d z / d logvar", atol_frac=1e-7)
    assert_close(Rb.grad.numpy(), Ra.grad.numpy(), 1e-6, "d z / d radius", atol_frac=1e-6)
