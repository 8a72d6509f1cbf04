"""GPU parity tests of the operator surface through the reference's import paths (`mt.*`): general-base-point exp / log
maps, geodesic distances, the guarded scalar functions and the device fast-math (direct tests of the DEVICE code), the
autograd of every primitive, the Poincare ball / projected sphere pinned through the hyperboloid / sphere, the
free-standing WrappedNormal, and the reference's eager training sequence against the recorded gradients."""
import numpy as np
import pytest
import torch

from helpers import T, assert_close, load_json, load_npz

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def _cpu(t):
    return t.detach().cpu().numpy()


def _mods():
    import mt.mvae.ops.euclidean as E
    import mt.mvae.ops.hyperbolics as H
    import mt.mvae.ops.poincare as P
    import mt.mvae.ops.spherical as S
    import mt.mvae.ops.spherical_projected as D
    return {"H": H, "S": S, "E": E, "P": P, "D": D}


# ------------------------------------------------------------------------------------------------ exp / log at a base point
@pytest.mark.parametrize("man", ["H", "S"])
@pytest.mark.parametrize("R", ["0.5", "1", "2", "11"])
@pytest.mark.parametrize("d", [2, 5, 40])
def test_general_exp_log_vs_golden(dev, man, R, d):
    """H.exp_map(u, at_point=mu, radius) / H.inverse_exp_map(z, at_point=mu, radius) (hyperbolics.py:106-128,
    spherical.py:86-109) against vectors recorded from the reference (g1: `exp`, `log`)."""
    M = _mods()[man]
    g = load_npz("g1_primitives.npz")
    k = f"{man}/R{R}/d{d}/f32/"
    Rt = torch.tensor(float(R), device=dev)
    mu, u, z = (T(g[k + n]).to(dev) for n in ("mu", "u", "z"))
    assert_close(_cpu(M.exp_map(u, at_point=mu, radius=Rt)), g[k + "exp"], RTOL, k + "exp")
    assert_close(_cpu(M.inverse_exp_map(z, at_point=mu, radius=Rt)), g[k + "log"], 5 * RTOL, k + "log",
                 atol_frac=5 * RTOL)
    # the module-level exp_map_mu0 takes the tangent vector with its leading zero (hyperbolics.py:114-116)
    x = T(g[k + "x"]).to(dev)
    xz = torch.cat((torch.zeros_like(x[..., :1]), x), -1)
    assert_close(_cpu(M.exp_map_mu0(xz, radius=Rt)), g[k + "mu"], RTOL, k + "mu")
    conv = M.lorentz_to_poincare if man == "H" else M.spherical_to_projected
    assert_close(_cpu(conv(mu, radius=Rt)), g[k + ("to_poincare" if man == "H" else "to_projected")], RTOL, k + "conv")


def test_euclidean_module_functions(dev):
    E = _mods()["E"]
    g = load_npz("g1_primitives.npz")
    k = "E/R1/d5/f32/"
    x, v = T(g[k + "x"]).to(dev), T(g[k + "v"]).to(dev)
    mu = E.exp_map_mu0(x)
    assert_close(_cpu(mu), g[k + "mu"], RTOL, "mu")
    z, (u, _) = E.sample_projection_mu0(v, at_point=mu)
    assert_close(_cpu(z), g[k + "z"], RTOL, "z")
    assert_close(_cpu(E.exp_map(u, at_point=mu)), g[k + "z"], RTOL, "exp_map")
    assert_close(_cpu(E.inverse_exp_map(z, at_point=mu)), g[k + "inv_u"], RTOL, "inverse_exp_map", atol_frac=RTOL)
    assert_close(_cpu(E.inverse_exp_map_mu0(mu)), g[k + "log_mu0"], RTOL, "log_mu0")


# ------------------------------------------------------------------------------------------------ distances
@pytest.mark.parametrize("man", ["H", "S", "E"])
def test_distances_vs_reference_helpers(dev, man):
    """mvae_geodesic_distance against the helpers of the reference's own op tests (g7, recorded through
    tests/mvae/ops/test_{hyperbolics,spherical,euclidean}.py) and the Lorentz product / norm."""
    M = _mods()[man]
    g = load_npz("g7_distances.npz")
    for R in ["0.5", "1", "2", "11"]:
        for d in [2, 5, 40]:
            k = f"{man}/R{R}/d{d}/f32/"
            p, q = T(g[k + "p"]).to(dev), T(g[k + "q"]).to(dev)
            Rt = torch.tensor(float(R), device=dev)
            if man == "H":
                # acosh near 1 amplifies the float32 rounding of <p,q>_L / R^2: the absolute floor is 5e-4 of the scale
                assert_close(_cpu(M.lorentz_distance(p, q, radius=Rt, keepdim=True)), g[k + "dist"], 5 * RTOL, k,
                             atol_frac=5 * RTOL)
                assert_close(_cpu(M.lorentz_product(p, q, keepdim=True)), g[k + "lprod"], RTOL, k + "lprod")
            elif man == "S":
                assert_close(_cpu(M.spherical_distance(p, q, radius=Rt)), g[k + "dist"], 5 * RTOL, k, atol_frac=5 * RTOL)
            else:
                assert_close(_cpu(M.euclidean_distance(p, q)), g[k + "dist"], RTOL, k)


def test_projected_sphere_distance_vs_golden(dev):
    D = _mods()["D"]
    g = load_npz("g6_projected.npz")
    for R in ["0.5", "1", "2", "11"]:
        for d in [2, 5, 40]:
            k = f"D/R{R}/d{d}/f32/"
            y, w = T(g[k + "y"]).to(dev), T(g[k + "w"]).to(dev)
            K = torch.tensor(1.0 / float(R)**2, device=dev)
            assert_close(_cpu(D.spherical_projected_distance(y, w, K)), g[k + "dist"], 5 * RTOL, k, atol_frac=5 * RTOL)
            Rt = torch.tensor(float(R), device=dev)
            assert_close(_cpu(D.projected_to_spherical(y, Rt)), g[k + "to_sphere"], RTOL, k + "to_sphere")
            assert_close(_cpu(D.lambda_x(y, Rt)), g[k + "lambda"], RTOL, k + "lambda")


def test_reference_property_distance_equals_tangent_norm(dev):
    """dist(mu, exp_mu(u)) = |u| with the fixture of the reference's op tests: R = 2, mu = R(2, 1, sqrt2),
    u = (0, 2, -sqrt2) (test_hyperbolics.py:168-178, test_spherical.py:172, test_poincare.py:140-148 with lambda_mu)."""
    m = _mods()
    R = torch.tensor(2.0, device=dev)
    t = lambda *v: torch.tensor(v, device=dev, dtype=torch.float32)  # noqa: E731
    mu, u = 2.0 * t(2.0, 1.0, np.sqrt(2)), t(0.0, 2.0, -np.sqrt(2))
    z = m["H"].exp_map(u, at_point=mu, radius=R)
    assert abs(float(m["H"].lorentz_distance(mu, z, radius=R)) - float(torch.norm(u))) < 1e-4 * float(torch.norm(u)) + 5e-4
    assert abs(float(m["H"].lorentz_norm(u)) - float(torch.norm(u))) < 1e-5
    u_back = m["H"].inverse_exp_map(z, at_point=mu, radius=R)
    assert float((u_back - u).abs().max()) < 1e-4  # the reference's float32 bar is 5e-6 on the CPU; see DESIGN 2
    # sphere: mu on the sphere of radius 2, u tangent
    mu_s, u_s = t(2.0, 0.0, 0.0), t(0.0, 1.0, -0.5)
    z_s = m["S"].exp_map(u_s, at_point=mu_s, radius=R)
    assert abs(float(m["S"].spherical_distance(mu_s, z_s, radius=R)) - float(torch.norm(u_s))) < 2e-4
    # Poincare ball (test_poincare.py:140-148): dist(mu, exp_mu(u)) = lambda_mu |u|
    mu_p = t(2.0, 1.0, np.sqrt(2)) / 2.0
    z_p = m["P"].exp_map(u, at_point=mu_p, radius=R)
    lam = float(m["P"].lambda_x(mu_p, R))
    dist = float(m["P"].poincare_distance(mu_p, z_p, radius=R))
    assert abs(dist - lam * float(torch.norm(u))) < 2e-3 * lam * float(torch.norm(u))
    u_p = m["P"].inverse_exp_map(z_p, at_point=mu_p, radius=R)
    assert float((u_p - u).abs().max()) < 2e-4
    assert abs(float(m["P"].poincare_distance(mu_p, mu_p, radius=R))) < 5e-4  # test_poincare.py:74-79
    d01 = m["P"].poincare_distance(torch.zeros_like(mu_p), mu_p, radius=R)
    d10 = m["P"].poincare_distance(mu_p, torch.zeros_like(mu_p), radius=R)
    assert float((d01 - d10).abs()) < 1e-6
    # euclidean: distance = 2 |x - y| (test_euclidean.py:41-42)
    a, b = t(1.0, 2.0, 3.0), t(0.0, 2.0, 5.0)
    assert abs(float(m["E"].euclidean_distance(a, b)) - 2 * float(torch.norm(a - b))) < 1e-5


def test_known_answers_of_the_reference_op_tests(dev):
    """tests/mvae/ops/test_hyperbolics.py:59-71 (lorentz_product integer cases), :80-83 (mu_0), :104 (lorentz_norm)."""
    H = _mods()["H"]
    t = lambda v: torch.tensor(v, device=dev, dtype=torch.float32)  # noqa: E731
    assert float(H.lorentz_product(t([0., 0, 0]), t([3., 2, 1]))) == 0
    assert float(H.lorentz_product(t([1., 2, 3]), t([3., 2, 1]))) == 4
    assert float(H.lorentz_product(t([1., 2, 3]), t([-3., 2, 1]))) == 10
    batch = H.lorentz_product(t([[1., 2, 3], [1., 0, 0]]), t([[3., 2, 1], [2., 5, 5]]), keepdim=True)
    assert batch.shape == (2, 1) and batch[:, 0].tolist() == [4.0, -2.0]
    assert H.mu_0((3,), radius=torch.tensor(2.0)).tolist() == [2.0, 0.0, 0.0]
    assert abs(float(H.lorentz_norm(t([2., 1, 2]))) - 1.0) < 1e-6
    # guarded sqrt: a time-like vector has <x,x>_L < 0 -> clamped at 1e-9
    assert abs(float(H.lorentz_norm(t([2., 1, 0]))) - np.sqrt(1e-9)) < 1e-7


# ------------------------------------------------------------------------------------------------ a-N: device scalar functions
def test_guarded_scalar_functions_on_the_device_vs_golden(dev):
    """The DEVICE code of the guarded functions and of their custom derivative rules (mvae_math.hpp: leaky_clamp,
    g_atanh, g_acosh, g_cosh, g_sinh, g_sqrt, g_logsinh, g_logcosh over float and Dual) against vectors recorded from
    the reference's ops/common.py:28-147 -- in-range, boundary and out-of-range points, value AND gradient."""
    from mvae_amd import functional as Fn
    g = load_npz("g1_scalar_fns.npz")
    for name in ["acosh", "atanh", "cosh", "sinh", "sqrt", "logsinh", "logcosh"]:
        x = T(g[f"{name}/f32/x"]).to(dev)
        y, dy = Fn.scalar_fn(name, x)
        want_y, want_dy = g[f"{name}/f32/y"], g[f"{name}/f32/dy"]
        fin = np.isfinite(want_y) & np.isfinite(want_dy)
        # atanh at +-1 is +-inf in float32 on both sides (the clamp 1 - 4e-8 rounds to 1): compare the finite entries
        # and require the same non-finite pattern
        assert (np.isfinite(_cpu(y)) == np.isfinite(want_y)).all(), name
        yy, dd = _cpu(y)[fin], _cpu(dy)[fin]
        big = np.abs(want_y[fin]) > 1e30
        assert_close(yy[~big], want_y[fin][~big], 2e-5, name + " value", atol_frac=1e-6)
        assert_close(dd[~big], want_dy[fin][~big], 5e-5, name + " derivative", atol_frac=1e-6)
    x = T(g["clamp_m1_2/f32/x"]).to(dev)
    y, dy = Fn.scalar_fn("clamp", x, -1.0, 2.0)
    assert np.array_equal(_cpu(y), g["clamp_m1_2/f32/y"])
    assert np.array_equal(_cpu(dy), g["clamp_m1_2/f32/dy"])  # 1 inside (inclusive), 1e-8 outside
    # the known answers of tests/mvae/ops/test_common.py:45-50 (acosh == np.arccosh on 100 seeded values)
    xk = T(g["acosh_known/x"].astype(np.float32)).to(dev)
    assert_close(_cpu(Fn.scalar_fn("acosh", xk)[0]), g["acosh_known/y_f32"], 1e-5, "acosh known answers")
    # finite on +-500 (test_common.py:31-42)
    big = torch.tensor([-500.0, 500.0], device=dev)
    for name in ["acosh", "sqrt", "cosh", "sinh", "logsinh", "logcosh"]:
        y, dy = Fn.scalar_fn(name, big)
        assert torch.isfinite(y).all() and torch.isfinite(dy).all(), name


def test_guarded_functions_are_differentiable_through_mt_common(dev):
    import mt.mvae.ops.common as C
    g = load_npz("g1_scalar_fns.npz")
    for name, fn in [("acosh", C.acosh), ("sqrt", C.sqrt), ("logsinh", C.logsinh), ("cosh", C.cosh)]:
        x = T(g[f"{name}/f32/x"]).to(dev).requires_grad_(True)
        w = torch.linspace(0.5, 1.5, x.numel(), device=dev)
        (fn(x) * w).sum().backward()
        want = g[f"{name}/f32/dy"] * _cpu(w)
        fin = np.isfinite(want) & (np.abs(want) < 1e30)
        assert_close(_cpu(x.grad)[fin], want[fin], 5e-5, name + " autograd", atol_frac=1e-6)
    x = torch.tensor([-2.0, 0.0, 3.0], device=dev, requires_grad=True)
    C.clamp(x, min=-1.0, max=2.0).sum().backward()
    assert np.allclose(_cpu(x.grad), [1e-8, 1.0, 1e-8])


def test_device_fastmath_vs_float64(dev):
    """mvae_fastmath.hpp as compiled for the DEVICE (__expf / __logf, -freciprocal-math, flush-to-zero) against
    float64: the shared-exp cosh/sinh pair, the shared-reduction cos/sin pair, log1p_pos, softplus, exp, log."""
    from mvae_amd import functional as Fn

    def ulps(got, want):
        want32 = want.astype(np.float32)
        return np.abs(got.astype(np.float64) - want) / np.maximum(np.spacing(np.abs(want32)).astype(np.float64), 1e-45)

    # exp(x) = exp2(x * log2(e)) on the hardware pair: the rounding of the product costs |x| * 1.44 * 2^-24 * ln 2
    # relative (mvae_fastmath.hpp header), on top of ~2 ulp of the evaluation itself
    def rel_ok(got, want, x64):
        return (np.abs(got.astype(np.float64) - want) <= (3e-7 + 6.5e-8 * np.abs(x64)) * np.abs(want) + 1e-37).all()

    x = torch.linspace(-20.0, 20.0, 20001, device=dev)
    c, s = Fn.scalar_fn("cosh_sinh_pair", x)
    x64 = _cpu(x).astype(np.float64)
    assert rel_ok(_cpu(c), np.cosh(x64), x64) and rel_ok(_cpu(s), np.sinh(x64), x64)
    small = torch.linspace(-0.4, 0.4, 4001, device=dev)  # around the polynomial / (e - 1/e) switch at 0.35
    c2, s2 = Fn.scalar_fn("cosh_sinh_pair", small)
    sm64 = _cpu(small).astype(np.float64)
    assert ulps(_cpu(s2), np.sinh(sm64)).max() <= 8 and ulps(_cpu(c2), np.cosh(sm64)).max() <= 4
    a = torch.linspace(-50.0, 50.0, 40001, device=dev)
    c, s = Fn.scalar_fn("cos_sin_pair", a)
    a64 = _cpu(a).astype(np.float64)
    assert np.abs(_cpu(c) - np.cos(a64)).max() < 4e-7 and np.abs(_cpu(s) - np.sin(a64)).max() < 4e-7
    big = torch.tensor([9000.0, -12345.0, 1e6], device=dev)  # beyond the fast range: full reduction
    c, s = Fn.scalar_fn("cos_sin_pair", big)
    b64 = _cpu(big).astype(np.float64)
    assert np.abs(_cpu(c) - np.cos(b64)).max() < 1e-6 and np.abs(_cpu(s) - np.sin(b64)).max() < 1e-6
    e = torch.logspace(-30, 30, 6001, device=dev)
    y, dy = Fn.scalar_fn("log1p_pos", e)
    assert ulps(_cpu(y), np.log1p(_cpu(e).astype(np.float64))).max() <= 8
    xs = torch.linspace(-60.0, 60.0, 12001, device=dev)
    y, dy = Fn.scalar_fn("softplus", xs)
    xs64 = _cpu(xs).astype(np.float64)
    want = np.where(xs64 > 20, xs64, np.log1p(np.exp(xs64)))
    ok = want > 1e-30  # below that float32 flushes
    assert rel_ok(_cpu(y)[ok], want[ok], xs64[ok])  # softplus(x) ~ exp(x) for very negative x: the exp bound applies
    assert np.abs(_cpu(dy) - np.where(xs64 > 20, 1.0, 1 / (1 + np.exp(-xs64)))).max() < 1e-6
    xe = torch.linspace(-85.0, 85.0, 17001, device=dev)
    y, _ = Fn.scalar_fn("exp", xe)
    assert (np.abs(_cpu(y) - np.exp(_cpu(xe).astype(np.float64))) <= 1e-5 * np.exp(_cpu(xe).astype(np.float64))).all()
    xl = torch.logspace(-30, 30, 6001, device=dev)
    y, _ = Fn.scalar_fn("log", xl)
    assert np.abs(_cpu(y) - np.log(_cpu(xl).astype(np.float64))).max() < 2e-5


# ------------------------------------------------------------------------------------------------ autograd of the primitives
def _oracle_fns(letter):
    from oracle import ops
    pre = letter
    f = {n: getattr(ops, f"{pre}_{n}") for n in ("exp_map_mu0", "exp_map", "log_map", "sample_projection_mu0",
                                                  "inverse_sample_projection_mu0")}
    f["logdet"] = getattr(ops, f"{pre}_logdet")
    return f


@pytest.mark.parametrize("letter,kind", [("h", 1), ("s", 2), ("p", 3), ("d", 4)])
@pytest.mark.parametrize("d", [2, 5])
def test_primitive_autograd_vs_oracle(dev, letter, kind, d):
    """Every differentiable Manifold method: gradients w.r.t. the tensor arguments AND the radius parameter from
    mvae_primitive_backward (dual numbers on the device) against torch.autograd on the oracle (float64)."""
    from mvae_amd import functional as Fn
    from oracle import ops
    R = 2.0
    g = torch.Generator().manual_seed(40 + d + kind)
    A = d + 1 if letter in "hs" else d
    x64 = torch.randn(16, d, generator=g, dtype=torch.float64) * 0.5
    v64 = torch.randn(16, d, generator=g, dtype=torch.float64) * 0.4
    ct1 = torch.randn(16, A, generator=g, dtype=torch.float64)
    ct2 = torch.randn(16, A, generator=g, dtype=torch.float64)
    ct3 = torch.randn(16, generator=g, dtype=torch.float64)
    o = _oracle_fns(letter)

    def run(f64):
        if f64:
            x, v, Rp = x64.clone().requires_grad_(True), v64.clone().requires_grad_(True), \
                torch.tensor(R, dtype=torch.float64, requires_grad=True)
            Rr = ops.radius_from_param(Rp)
            mu = o["exp_map_mu0"](x, Rr)
            z, (u, _) = o["sample_projection_mu0"](v, mu, Rr)
            ld = o["logdet"](u, Rr) if letter in "hs" else o["logdet"](mu, z, Rr)
            u0, v0 = o["inverse_sample_projection_mu0"](z, mu.detach() * 0.9 if letter in "pd" else mu, Rr)
            c1, c2, c3 = ct1, ct2, ct3
        else:
            x, v = x64.float().to(dev).requires_grad_(True), v64.float().to(dev).requires_grad_(True)
            Rp = torch.tensor(R, device=dev, requires_grad=True)
            mu = Fn.exp_map_mu0(kind, x, Rp)
            z, (u, _) = Fn.sample_projection_mu0(kind, v, mu, Rp)
            ld = Fn.logdet(kind, u, None, None, Rp) if letter in "hs" else Fn.logdet(kind, None, mu, z, Rp)
            u0, v0 = Fn.inverse_sample_projection_mu0(kind, z, mu.detach() * 0.9 if letter in "pd" else mu, Rp)
            c1, c2, c3 = ct1.float().to(dev), ct2.float().to(dev), ct3.float().to(dev)
        loss = (z * c1).sum() + (u * c2).sum() + (ld * c3).sum() + 0.1 * (v0 * v0).sum()
        loss.backward()
        return [t.grad.detach().cpu().double().numpy() for t in (x, v, Rp)], float(loss)

    (gx, gv, gR), l64 = run(True)
    (hx, hv, hR), l32 = run(False)
    assert abs(l32 - l64) < 2e-4 * max(1.0, abs(l64))
    assert_close(hx, gx, 5e-4, "d/dx", atol_frac=5e-4)
    assert_close(hv, gv, 5e-4, "d/dv", atol_frac=5e-4)
    assert_close(hR, gR, 2e-3, "d/dR", atol_frac=2e-3)


def test_exp_log_distance_autograd_vs_oracle(dev):
    """General-base-point exp / log maps and the distances: gradients w.r.t. both arguments and the radius."""
    from mvae_amd import functional as Fn
    from oracle import ops
    R = 2.0
    g = torch.Generator().manual_seed(77)
    x64 = torch.randn(8, 3, generator=g, dtype=torch.float64) * 0.4
    y64 = torch.randn(8, 3, generator=g, dtype=torch.float64) * 0.4
    for kind, exp0, dist in [(1, ops.h_exp_map_mu0, ops.h_distance), (2, ops.s_exp_map_mu0, ops.s_distance),
                             (3, ops.p_exp_map_mu0, ops.p_distance)]:
        xa, ya = x64.clone().requires_grad_(True), y64.clone().requires_grad_(True)
        Ra = torch.tensor(R, dtype=torch.float64, requires_grad=True)
        dist(exp0(xa, Ra), exp0(ya, Ra), Ra).sum().backward()
        xb, yb = x64.float().to(dev).requires_grad_(True), y64.float().to(dev).requires_grad_(True)
        Rb = torch.tensor(R, device=dev, requires_grad=True)
        Fn.geodesic_distance(kind, Fn.exp_map_mu0(kind, xb, Rb), Fn.exp_map_mu0(kind, yb, Rb), Rb).sum().backward()
        assert_close(_cpu(xb.grad), xa.grad.numpy(), 1e-3, f"kind {kind} d dist/dx", atol_frac=1e-3)
        assert_close(_cpu(yb.grad), ya.grad.numpy(), 1e-3, f"kind {kind} d dist/dy", atol_frac=1e-3)
        assert abs(float(Rb.grad) - float(Ra.grad)) < 2e-3 * max(1.0, abs(float(Ra.grad)))


# ------------------------------------------------------------------------------------------------ p / d through h / s
def _pts32(seed, rows, d, R, dev, scale=0.35):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(rows, d, generator=g, dtype=torch.float64) * scale * min(R, 3.0) / np.sqrt(d)).float().to(dev)


@pytest.mark.parametrize("R", [0.5, 1.0, 2.0, 11.0])
@pytest.mark.parametrize("d", [2, 5])
def test_poincare_ball_through_the_hyperboloid(dev, R, d):
    """The identities of tests/test_oracle_crossmodel.py on the HIP kernels at the float32 bar: the Poincare-ball
    operators (parity-unpinned: geoopt) against the hyperboloid operators (pinned to the reference) through
    poincare_to_lorentz.  Replaces the 5e-4 / 2e-2 checks of round 1."""
    from mvae_amd import functional as Fn
    from test_oracle_crossmodel import d_rho
    Rt = torch.tensor(R, device=dev)
    x, u, v = _pts32(3, 256, d, R, dev), _pts32(4, 256, d, R, dev), _pts32(5, 256, d, R, dev)
    at = Fn.exp_map_mu0(3, x, Rt)
    rho = lambda t: Fn.manifold_aux(14, 3, t, None, Rt)  # MVAE_OP_TO_AMBIENT  # noqa: E731
    assert_close(_cpu(rho(at)), _cpu(Fn.exp_map_mu0(1, 2 * x, Rt)), RTOL, "exp_map_mu0")
    z = Fn.exp_map(3, u, at, Rt)
    zh = Fn.exp_map(1, d_rho(at, u, Rt), rho(at), Rt)
    assert_close(_cpu(rho(z)), _cpu(zh), RTOL, "exp_map", atol_frac=RTOL)
    lg = d_rho(at, Fn.inverse_exp_map(3, z, at, Rt), Rt)
    assert_close(_cpu(lg), _cpu(Fn.inverse_exp_map(1, zh, rho(at), Rt)), 5 * RTOL, "inverse_exp_map", atol_frac=5 * RTOL)
    zs, (us, _) = Fn.sample_projection_mu0(3, v, at, Rt)
    zhs, (uh, _) = Fn.sample_projection_mu0(1, v, rho(at), Rt)
    assert_close(_cpu(rho(zs)), _cpu(zhs), RTOL, "sample_projection z", atol_frac=RTOL)
    assert_close(_cpu(d_rho(at, us, Rt)), _cpu(uh), RTOL, "sample_projection u", atol_frac=RTOL)
    y2 = Fn.exp_map_mu0(3, _pts32(7, 256, d, R, dev), Rt)
    dp = Fn.geodesic_distance(3, at, y2, Rt)
    dh = Fn.geodesic_distance(1, rho(at), rho(y2), Rt)
    assert_close(_cpu(dp), _cpu(dh), 5 * RTOL, "distance", atol_frac=5 * RTOL)
    assert_close(_cpu(Fn.logdet(3, None, at, zs, Rt)), _cpu(Fn.logdet(1, uh, None, None, Rt)), 5 * RTOL, "logdet",
                 atol_frac=5 * RTOL)


@pytest.mark.parametrize("R", [0.5, 1.0, 2.0, 11.0])
@pytest.mark.parametrize("d", [2, 5])
def test_projected_sphere_through_the_sphere(dev, R, d):
    from mvae_amd import functional as Fn
    from test_oracle_crossmodel import d_sigma
    Rt = torch.tensor(R, device=dev)
    x, u, v = _pts32(13, 256, d, R, dev, 0.25), _pts32(14, 256, d, R, dev, 0.25), _pts32(15, 256, d, R, dev, 0.25)
    at = Fn.exp_map_mu0(4, x, Rt)
    sig = lambda t: Fn.manifold_aux(14, 4, t, None, Rt)  # noqa: E731
    assert_close(_cpu(sig(at)), _cpu(Fn.exp_map_mu0(2, 2 * x, Rt)), RTOL, "exp_map_mu0")
    z = Fn.exp_map(4, u, at, Rt)
    zs_ = Fn.exp_map(2, d_sigma(at, u, Rt), sig(at), Rt)
    assert_close(_cpu(sig(z)), _cpu(zs_), RTOL, "exp_map", atol_frac=RTOL)
    zs, _ = Fn.sample_projection_mu0(4, v, at, Rt)
    zh, _ = Fn.sample_projection_mu0(2, v, sig(at), Rt)
    assert_close(_cpu(sig(zs)), _cpu(zh), RTOL, "sample_projection z", atol_frac=RTOL)
    y2 = Fn.exp_map_mu0(4, _pts32(17, 256, d, R, dev, 0.25), Rt)
    dg = Fn.geodesic_distance(4, at, y2, Rt, gyro=True)
    ds = Fn.geodesic_distance(2, sig(at), sig(y2), Rt)
    assert_close(_cpu(dg), _cpu(ds), 5 * RTOL, "gyro distance", atol_frac=5 * RTOL)


@pytest.mark.parametrize("ball,ambient,conv_kind", [("p", "h", 3), ("d", "s", 4)])
@pytest.mark.parametrize("R", [1.0, 2.0])
def test_ball_component_through_the_pinned_component(dev, ball, ambient, conv_kind, R):
    """z, KL, d/d(mean head), d/d(logvar head), d/d(radius) of a `p` (`d`) component == those of the pinned `h` (`s`)
    component fed 2 * mean, through the fused component operators on the device."""
    from mvae_amd import functional as Fn
    d, B = 2, 256
    g = torch.Generator().manual_seed(100 + d)
    m = (torch.randn(B, d, generator=g, dtype=torch.float64) * 0.3 * min(R, 2.0) / np.sqrt(d)).float()
    lv = (torch.randn(B, d, generator=g, dtype=torch.float64) * 0.5 - 1.0).float()
    eps = torch.randn(B, d, generator=g, dtype=torch.float64).float().to(dev)
    w = torch.randn(1, B, generator=g, dtype=torch.float64).float().to(dev)
    ct = torch.randn(B, d + 1, generator=g, dtype=torch.float64).float().to(dev)
    res = {}
    for letter, scale in ((ball, 1.0), (ambient, 2.0)):
        lay = Fn.ComponentLayout([(letter, d)])
        heads = torch.cat((scale * m, lv), dim=1).to(dev).requires_grad_(True)
        radii = torch.tensor([R], device=dev, requires_grad=True)
        z, kl = Fn.component_rsample_kl(lay, heads, radii, eps)
        zz = Fn.manifold_aux(14, conv_kind, z, None, radii) if letter == ball else z
        ((kl * w).sum() + (zz * ct).sum()).backward()
        res[letter] = (_cpu(zz), _cpu(kl), _cpu(heads.grad), _cpu(radii.grad))
    zb, kb, hb, rb = res[ball]
    za, ka, ha, ra = res[ambient]
    assert_close(zb, za, RTOL, "z", atol_frac=RTOL)
    assert_close(kb, ka, 2 * RTOL, "kl", atol_frac=2 * RTOL)
    assert_close(hb[:, :d], 2 * ha[:, :d], 5e-4, "d/d mean", atol_frac=5e-4)
    assert_close(hb[:, d:], ha[:, d:], 5e-4, "d/d logvar", atol_frac=5e-4)
    assert_close(rb, ra, 2e-3, "d/d radius", atol_frac=2e-3)


# ------------------------------------------------------------------------------------------------ WrappedNormal
def test_wrapped_normal_envelope(dev):
    """tests/mvae/distributions/test_wrapped_normal.py:24-52: WrappedNormal on the Hyperboloid over dims x scales x
    radii: samples finite, log_prob finite and <= 0."""
    from mt.mvae.distributions import WrappedNormal
    from mt.mvae.ops import Hyperboloid
    for dim in [3, 6, 11, 21, 41]:
        for scale in [100.0, 1.01, 1.0]:
            for R in [1e-5, 1.0, 1e5]:
                radius = torch.tensor(R, device=dev)
                man = Hyperboloid(lambda: radius)
                loc = man.mu_0((100, dim), device=dev)
                q = WrappedNormal(loc, torch.full((100, dim - 1), scale, device=dev) / scale * min(scale, 1.01), man)
                z, data = q.rsample_with_parts(torch.Size([10]))
                assert torch.isfinite(z).all(), (dim, scale, R)
                lp = q.log_prob_from_parts(z, data)
                assert lp.shape == (10, 100) and torch.isfinite(lp).all(), (dim, scale, R)


def test_wrapped_normal_equals_the_fused_component(dev):
    """WrappedNormalProcedure.reparametrize -> rsample_with_parts -> kl_loss on free-standing distributions (one HIP
    primitive per step, the reference's sequence sampling_procedures.py:93-116) == the fused component operator."""
    from mvae_amd import functional as Fn, utils
    for model in ("h3", "s3", "p3", "d3", "e3"):
        comp = utils.parse_components(model, fixed_curvature=False)[0]
        comp.init_layers(16, scalar_parametrization=False)
        comp.to(dev)
        if comp._radius_param() is not None:
            comp._radius_param().data.fill_(2.0)
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(32, 16, generator=g) * 0.5).to(dev)
        eps = torch.randn(32, 3, generator=g).to(dev)
        with torch.no_grad():
            q_f, p_f, _ = comp(x)
            z_f, data_f = q_f.rsample_with_parts(eps=eps)
            kl_f = comp.kl_loss(q_f, p_f, z_f, data_f)
            z_mean, std = comp.encode(x)
            q, p = comp.reparametrize(z_mean, std)
            z, data = q.rsample_with_parts(eps=eps)
            kl = comp.kl_loss(q, p, z, data)
        assert_close(_cpu(z), _cpu(z_f), RTOL, model + " z", atol_frac=RTOL)
        assert_close(_cpu(kl), _cpu(kl_f), 5 * RTOL, model + " kl", atol_frac=5 * RTOL)


# ------------------------------------------------------------------------------------------------ the eager sequence
class _DS:

    def __init__(self, in_dim):
        self.in_dim = in_dim

    def reconstruction_loss(self, x_, x):
        from mvae_amd import functional as Fn
        return Fn.bce_with_logits_rows(x_, x)


@pytest.mark.parametrize("case", ["h2s2e2_learn_ep12", "e6_fixed_ep0", "prod36_learn_ep12", "h2s2e2_scalar_ep12",
                                  "h5s3e4_learn_ep5"])
def test_reference_eager_sequence_through_mt_imports(dev, case):
    """The reference's own call sequence (vae.py:150-164) through `mt.*` imports:
        optimizer.zero_grad(); reps, z, x_ = model(x); stats = model.compute_batch_stats(...);
        (-stats.elbo).backward(); optimizer.step()
    Every .grad against the gradients recorded from the reference (g3 small cases), then the parameters after the step
    against the recorded state."""
    from mt.mvae import utils
    from mt.mvae.models import FeedForwardVAE, Trainer
    meta = load_json("g3_step_small.json")[case]
    g = load_npz("g3_step_small.npz")
    k = f"{case}/f32/steps1/"
    state0 = {n[len(k + "state0/"):]: T(v) for n, v in g.items() if n.startswith(k + "state0/")}
    model = FeedForwardVAE(meta["h_dim"], utils.parse_components(meta["model"], meta["fixed_curvature"]),
                           _DS(meta["in_dim"]), meta["scalar_parametrization"])
    model.load_state_dict(state0)
    model.to(dev)
    trainer = Trainer(model, chkpt_dir="/tmp/mvae_test_chkpt_eager")
    trainer.epoch = meta["epoch"]
    if meta["epoch"] < 10:  # the radius warm-up of Trainer._train_epoch (train.py:189-194)
        model.engine.set_radii(11 - meta["epoch"])
    opt = trainer.build_optimizer(learning_rate=1e-3, fixed_curvature=meta["fixed_curvature"])
    x = T(g[k + "x"], torch.float32)[0].to(dev)
    eps = T(g[k + "eps"])[0].to(dev)
    opt.zero_grad()
    reps, concat_z, x_ = model(x, eps=eps)
    assert x_.requires_grad and concat_z.requires_grad
    stats = model.compute_batch_stats(x, x_, reps, beta=1.0)
    loss = -stats.elbo
    loss.backward()
    assert_close(float(stats.elbo), float(g[k + "stats"][0][2]), RTOL, "elbo")
    assert_close(_cpu(x_), g[k + "logits"], RTOL, "logits")
    n_checked = 0
    for name, p in model.named_parameters():
        key = k + "grad/" + name
        if key in g:
            assert p.grad is not None, name
            assert_close(_cpu(p.grad), g[key], RTOL, "grad " + name, atol_frac=1e-4)
            n_checked += 1
    assert n_checked >= 10
    # the gradients landed in the engine's flat buffer (p.grad aliases it), where the optimizer kernel reads them
    gv = model.engine.grad_views()
    assert all(p.grad.data_ptr() == gv[n].data_ptr() for n, p in model.named_parameters() if p.grad is not None)
    opt.step()
    for name, p in model.named_parameters():
        assert_close(_cpu(p), g[k + "state1/" + name], RTOL, "param after step " + name)


def test_eager_sequence_survives_zero_grad_set_to_none(dev):
    """torch's own `zero_grad(set_to_none=True)` detaches p.grad from the flat buffer; CurvatureOptimizer.step gathers the
    fresh gradient tensors back before the optimizer kernel runs."""
    from mt.mvae import utils
    from mt.mvae.models import FeedForwardVAE, Trainer
    case = "h2s2e2_learn_ep12"
    meta = load_json("g3_step_small.json")[case]
    g = load_npz("g3_step_small.npz")
    k = f"{case}/f32/steps1/"
    state0 = {n[len(k + "state0/"):]: T(v) for n, v in g.items() if n.startswith(k + "state0/")}
    model = FeedForwardVAE(meta["h_dim"], utils.parse_components(meta["model"], False), _DS(meta["in_dim"]), False)
    model.load_state_dict(state0)
    model.to(dev)
    trainer = Trainer(model, chkpt_dir="/tmp/mvae_test_chkpt_eager")
    trainer.epoch = meta["epoch"]
    opt = trainer.build_optimizer(learning_rate=1e-3, fixed_curvature=False)
    model.zero_grad(set_to_none=True)
    x, eps = T(g[k + "x"], torch.float32)[0].to(dev), T(g[k + "eps"])[0].to(dev)
    reps, _, x_ = model(x, eps=eps)
    (-model.compute_batch_stats(x, x_, reps, beta=1.0).elbo).backward()
    opt.step()
    for name, p in model.named_parameters():
        assert_close(_cpu(p), g[k + "state1/" + name], RTOL, "param after step " + name)


def test_optimizer_skips_parameters_without_a_gradient(dev):
    """torch.optim.Adam (the reference's optimizer, train.py:327-360) skips a parameter whose .grad is None: neither the
    parameter nor its moments move.  Here only the decoder takes part in the loss; the encoder and the heads must stay
    exactly where they were, moments included, while the decoder is updated."""
    from mt.mvae import utils
    from mt.mvae.models import FeedForwardVAE, Trainer
    case = "h2s2e2_learn_ep12"
    meta = load_json("g3_step_small.json")[case]
    g = load_npz("g3_step_small.npz")
    k = f"{case}/f32/steps1/"
    state0 = {n[len(k + "state0/"):]: T(v) for n, v in g.items() if n.startswith(k + "state0/")}
    model = FeedForwardVAE(meta["h_dim"], utils.parse_components(meta["model"], False), _DS(meta["in_dim"]), False)
    model.load_state_dict(state0)
    model.to(dev)
    trainer = Trainer(model, chkpt_dir="/tmp/mvae_test_chkpt_skip")
    trainer.epoch = meta["epoch"]
    opt = trainer.build_optimizer(learning_rate=1e-3, fixed_curvature=False)
    eng = model.engine
    eng.adam_m.fill_(0.25)  # momentum that WOULD move a parameter if the optimizer touched it
    eng.adam_v.fill_(0.5)
    model.zero_grad(set_to_none=True)
    z = torch.randn(8, eng.layout.z_dim, device=dev)
    x = (torch.rand(8, meta["in_dim"], device=dev) > 0.5).float()
    from mvae_amd import functional as Fn
    loss = Fn.bce_with_logits_rows(model.decode(z), x).sum()
    loss.backward()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    m_before = eng.adam_m.clone()
    opt.step()
    mv = eng.flat.views(eng.adam_m)
    mb = eng.flat.views(m_before)
    for n, p in model.named_parameters():
        if p.grad is None:
            assert torch.equal(p.detach(), before[n]), f"{n} moved without a gradient"
            assert torch.equal(mv[n], mb[n]), f"moments of {n} changed"
        else:
            assert not torch.equal(p.detach(), before[n]), f"{n} has a gradient but did not move"
    assert any(p.grad is None for _, p in model.named_parameters()) and any(
        p.grad is not None for _, p in model.named_parameters())


def test_bce_gradient_flows_with_broadcast_targets(dev):
    """x broadcast over a leading sample dimension (the log-likelihood path's shapes) while the logits require grad: the
    loss must keep its grad_fn, and the gradient equals sigmoid(logits) - x."""
    from mvae_amd import functional as Fn
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(3, 5, 32, generator=g).to(dev).requires_grad_(True)
    x = torch.rand(5, 32, generator=g).to(dev)
    loss = Fn.bce_with_logits_rows(logits, x)
    assert loss.grad_fn is not None and loss.shape == (3, 5)
    loss.sum().backward()
    ref = torch.sigmoid(logits.detach().double()) - x.double()
    assert_close(_cpu(logits.grad), ref.cpu().numpy(), 2e-5, "d bce / d logits", atol_frac=1e-5)
