"""ORACLE (test infrastructure) -- manifold primitives and guarded scalar functions, CPU PyTorch.

Restates mt/mvae/ops/{common,manifold,hyperbolics,spherical,euclidean,poincare}.py of the reference.
All citations are relative to /root/reference.  Tensors are [..., A] with coordinates in the last dim.
"""
import math
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

EPS = 1e-8  # common.py:21
MAX_NORM = 85.0  # common.py:22
LN_2 = math.log(2.0)

EUCLIDEAN, HYPERBOLOID, SPHERE, POINCARE, PROJ_SPHERE, UNIVERSAL = 0, 1, 2, 3, 4, 5
KIND_OF_LETTER = {"e": EUCLIDEAN, "h": HYPERBOLOID, "s": SPHERE, "p": POINCARE, "d": PROJ_SPHERE, "u": UNIVERSAL}


# --------------------------------------------------------------------------- guarded scalar functions
class _LeakyClamp(torch.autograd.Function):
    """common.py:28-39 -- forward hard clamp; backward passes g inside [min,max] (inclusive), g*1e-8 outside."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.save_for_backward((x >= lo) & (x <= hi))
        return torch.clamp(x, min=lo, max=hi)

    @staticmethod
    def backward(ctx, g):
        (inside,) = ctx.saved_tensors
        w = inside.to(g.dtype)
        return g * w + g * (1 - w) * EPS, None, None


def leaky_clamp(x: Tensor, lo: float = float("-inf"), hi: float = float("inf")) -> Tensor:
    return _LeakyClamp.apply(x, lo, hi)


class _Atanh(torch.autograd.Function):
    """common.py:46-63 -- clamp to +-(1-4e-8), 0.5*(log(1+x)-log(1-x)); backward g/(1-x^2) on the clamped x."""

    @staticmethod
    def forward(ctx, x):
        xc = torch.clamp(x, min=-1.0 + 4 * EPS, max=1.0 - 4 * EPS)
        ctx.save_for_backward(xc)
        return (torch.log(1 + xc) - torch.log(1 - xc)) * 0.5

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        return g / (1 - xc**2)


def atanh(x: Tensor) -> Tensor:
    return _Atanh.apply(x)


class _Acosh(torch.autograd.Function):
    """common.py:76-94 -- x<-max(x,1+1e-8); z=sqrt(max(x^2-1,1e-9)); log(x+z); backward g/z."""

    @staticmethod
    def forward(ctx, x):
        xc = torch.clamp(x, min=1 + EPS)
        z = torch.sqrt(torch.clamp(xc * xc - 1.0, min=1e-9))
        ctx.save_for_backward(z)
        return torch.log(xc + z)

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        return g / z


def acosh(x: Tensor) -> Tensor:
    return _Acosh.apply(x)


def cosh(x: Tensor) -> Tensor:  # common.py:107-109
    return torch.cosh(leaky_clamp(x, -MAX_NORM, MAX_NORM))


def sinh(x: Tensor) -> Tensor:  # common.py:112-114
    return torch.sinh(leaky_clamp(x, -MAX_NORM, MAX_NORM))


def sqrt(x: Tensor) -> Tensor:  # common.py:117-119
    return torch.sqrt(leaky_clamp(x, lo=1e-9))


def _signed_logsumexp2(a: Tensor, b: Tensor, sign_b: float) -> Tensor:
    """common.py:139-147 specialised to two terms with signs (+1, sign_b): m + log(clamp(e^(a-m) + s e^(b-m), 1e-8)).
    `m` is a max over the pair: its gradient goes to the arg-max element, as torch.max does."""
    stacked = torch.stack((a, b), dim=-1)
    m, _ = torch.max(stacked, dim=-1, keepdim=True)
    signs = torch.tensor([1.0, sign_b], dtype=a.dtype)
    s = torch.sum(signs * torch.exp(stacked - m), dim=-1)
    return m.squeeze(-1) + torch.log(leaky_clamp(s, lo=EPS))


def logsinh(x: Tensor) -> Tensor:  # common.py:122-128
    return x + _signed_logsumexp2(torch.zeros_like(x), -2.0 * x, -1.0) - LN_2


def logcosh(x: Tensor) -> Tensor:  # common.py:131-136 (plain torch.logsumexp there: no clamp)
    return x + torch.logsumexp(torch.stack((torch.zeros_like(x), -2.0 * x), dim=-1), dim=-1) - LN_2


def radius_from_param(p: Tensor) -> Tensor:  # manifold.py:73-75
    return torch.clamp(torch.relu(p), min=1e-8, max=1e8)


def _prepend_zero(x: Tensor) -> Tensor:  # common.py:156-158
    return torch.cat((torch.zeros_like(x[..., :1]), x), dim=-1)


# --------------------------------------------------------------------------- hyperboloid (hyperbolics.py)
def lorentz_product(x: Tensor, y: Tensor, keepdim: bool = False) -> Tensor:  # hyperbolics.py:72-78
    m = x * y
    if keepdim:
        return torch.sum(m, dim=-1, keepdim=True) - 2 * m[..., 0:1]
    return torch.sum(m, dim=-1) - 2 * m[..., 0]


def lorentz_norm(x: Tensor, keepdim: bool = False) -> Tensor:  # hyperbolics.py:81-84
    return sqrt(lorentz_product(x, x, keepdim=keepdim))


def h_mu0(shape, R: Tensor, dtype=None) -> Tensor:  # hyperbolics.py:68-69
    e0 = torch.zeros(shape, dtype=dtype or R.dtype)
    e0[..., 0] = 1
    return e0 * R


def h_exp_map_mu0(x: Tensor, R: Tensor) -> Tensor:
    """hyperbolics.py:28-29,114-121; `x` is the true-dim tangent vector (the zero coordinate is implicit)."""
    n = torch.norm(x, p=2, dim=-1, keepdim=True) / R
    direction = F.normalize(x, p=2, dim=-1) * R
    return torch.cat((cosh(n) * R, sinh(n) * direction), dim=-1)


def h_pt_mu0(x: Tensor, dst: Tensor, R: Tensor) -> Tensor:  # hyperbolics.py:87-93
    coef = lorentz_product(dst, x, keepdim=True) / (R * (R + dst[..., 0:1]))
    right = torch.cat((dst[..., 0:1] + R, dst[..., 1:]), dim=-1)
    return x + coef * right


def h_inv_pt_mu0(x: Tensor, src: Tensor, R: Tensor) -> Tensor:  # hyperbolics.py:96-103
    coef = -x[..., 0:1] / (R + src[..., 0:1])
    right = torch.cat((src[..., 0:1] + R, src[..., 1:]), dim=-1)
    return x + coef * right


def h_exp_map(x: Tensor, at: Tensor, R: Tensor) -> Tensor:  # hyperbolics.py:106-111
    n = lorentz_norm(x, keepdim=True) / R
    return cosh(n) * at + sinh(n) * (x / n)


def h_log_map(x: Tensor, at: Tensor, R: Tensor) -> Tensor:  # hyperbolics.py:124-128
    alpha = -lorentz_product(at, x, keepdim=True) / (R**2)
    return acosh(alpha) / sqrt(alpha**2 - 1) * (x - alpha * at)


def h_log_map_mu0(x: Tensor, R: Tensor) -> Tensor:  # hyperbolics.py:131-135
    alpha = x[..., 0:1] / R
    coef = acosh(alpha) / sqrt(alpha**2 - 1.0)
    return coef * torch.cat((x[..., 0:1] - alpha * R, x[..., 1:]), dim=-1)


def h_sample_projection_mu0(v: Tensor, at: Tensor, R: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:
    u = h_pt_mu0(_prepend_zero(v), at, R)  # hyperbolics.py:138-142
    return h_exp_map(u, at, R), (u, v)


def h_inverse_sample_projection_mu0(z: Tensor, at: Tensor, R: Tensor) -> Tuple[Tensor, Tensor]:
    u = h_log_map(z, at, R)  # hyperbolics.py:145-148
    return u, h_inv_pt_mu0(u, at, R)[..., 1:]


def h_logdet(u: Tensor, R: Tensor) -> Tensor:  # hyperbolics.py:58-65
    r = lorentz_norm(u) / R
    n = u.shape[-1] - 1
    return (n - 1) * (torch.log(R) + logsinh(r) - torch.log(r))


def lorentz_to_poincare(x: Tensor, R: Tensor) -> Tensor:  # hyperbolics.py:151-152
    return R * x[..., 1:] / (R + x[..., 0:1])


# --------------------------------------------------------------------------- sphere (spherical.py)
def s_exp_map_mu0(x: Tensor, R: Tensor) -> Tensor:  # spherical.py:28-29,94-101
    n = torch.norm(x, p=2, dim=-1, keepdim=True) / R
    direction = F.normalize(x, p=2, dim=-1) * R
    return torch.cat((torch.cos(n) * R, torch.sin(n) * direction), dim=-1)


def s_pt_mu0(v: Tensor, dst: Tensor, R: Tensor) -> Tensor:  # spherical.py:74-77
    coef = torch.sum(dst * v, dim=-1, keepdim=True) / (R * (R + dst[..., 0:1]))
    right = torch.cat((dst[..., 0:1] + R, dst[..., 1:]), dim=-1)
    return v - coef * right


def s_inv_pt_mu0(x: Tensor, src: Tensor, R: Tensor) -> Tensor:  # spherical.py:80-83
    coef = x[..., 0:1] / (R + src[..., 0:1])
    right = torch.cat((src[..., 0:1] + R, src[..., 1:]), dim=-1)
    return x - coef * right


def s_exp_map(x: Tensor, at: Tensor, R: Tensor) -> Tensor:  # spherical.py:86-91 (no clamp on the norm)
    n = torch.norm(x, p=2, dim=-1, keepdim=True) / R
    return torch.cos(n) * at + torch.sin(n) * (x / n)


def s_log_map(x: Tensor, at: Tensor, R: Tensor) -> Tensor:  # spherical.py:104-109 (hard clamp on alpha)
    alpha = torch.sum(at * x, dim=-1, keepdim=True) / (R**2)
    coef = torch.acos(torch.clamp(alpha, min=-1.0, max=1.0)) / sqrt(1.0 - alpha**2)
    return coef * (x - alpha * at)


def s_log_map_mu0(x: Tensor, R: Tensor) -> Tensor:  # spherical.py:112-116
    alpha = x[..., 0:1] / R
    coef = torch.acos(torch.clamp(alpha, min=-1.0, max=1.0)) / sqrt(1.0 - alpha**2)
    return coef * torch.cat((x[..., 0:1] - alpha * R, x[..., 1:]), dim=-1)


def s_sample_projection_mu0(v: Tensor, at: Tensor, R: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:
    u = s_pt_mu0(_prepend_zero(v), at, R)  # spherical.py:119-123
    return s_exp_map(u, at, R), (u, v)


def s_inverse_sample_projection_mu0(z: Tensor, at: Tensor, R: Tensor) -> Tuple[Tensor, Tensor]:
    u = s_log_map(z, at, R)  # spherical.py:126-129
    return u, s_inv_pt_mu0(u, at, R)[..., 1:]


def s_logdet(u: Tensor, R: Tensor) -> Tensor:  # spherical.py:58-67 (hard clamps at 1e-5)
    r = torch.norm(u, dim=-1, p=2) / R
    n = u.shape[-1] - 1
    return (n - 1) * (torch.log(R) + torch.log(torch.abs(torch.sin(r)).clamp(min=1e-5)) -
                      torch.log(r.clamp(min=1e-5)))


def spherical_to_projected(x: Tensor, R: Tensor) -> Tensor:  # spherical.py:132-133
    return R * x[..., 1:] / (R + x[..., 0:1])


# --------------------------------------------------------------------------- euclidean (euclidean.py)
def e_exp_map_mu0(x: Tensor) -> Tensor:  # euclidean.py:78-79 (the factor 1/2 is the reference's, pinned by its tests)
    return x / 2


def e_log_map_mu0(x: Tensor) -> Tensor:  # euclidean.py:86-87
    return 2 * x


def e_sample_projection_mu0(v: Tensor, at: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:
    return at + v / 2, (v, v)  # euclidean.py:90-93 (never used by the Euclidean *Normal* procedure)


def e_inverse_sample_projection_mu0(z: Tensor, at: Tensor) -> Tuple[Tensor, Tensor]:
    u = 2 * (z - at)  # euclidean.py:96-99
    return u, u


# --------------------------------------------------------------------------- Poincare ball  (PARITY UNPINNED)
# poincare.py delegates to geoopt==0.1.0 (absent).  Formulas: Ganea et al. 2018, guards as in geoopt 0.1.0 as best
# known: MIN_NORM=1e-15 on norms, tanh argument clamped to +-15, mobius_add denominator clamp_min(MIN_NORM).
# The denominator guard is anchored on the reference's own tests: its exp/log round trips to atol 5e-6
# (tests/mvae/ops/test_poincare.py:140-211, test_spherical_projected.py:202-270) hold with clamp_min(1e-15) and fail
# by two orders of magnitude with the "+ 1e-5" variant some geoopt versions carried; the commented-out restatement in
# spherical_projected.py:108-112 shows the same `denom.clamp(min=MIN_NORM)` form.
P_MIN_NORM = 1e-15


def _p_c(R: Tensor) -> Tensor:  # poincare.py:108-109
    return 1 / R**2


def _p_tanh(x: Tensor) -> Tensor:
    return torch.tanh(torch.clamp(x, -15.0, 15.0))


class _PArtanh(torch.autograd.Function):
    """geoopt 0.1.0 artanh: clamp to +-(1-1e-5), backward g/(1-x^2)."""

    @staticmethod
    def forward(ctx, x):
        xc = torch.clamp(x, -1 + 1e-5, 1 - 1e-5)
        ctx.save_for_backward(xc)
        return (torch.log(1 + xc) - torch.log(1 - xc)) * 0.5

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        return g / (1 - xc**2)


def p_lambda_x(x: Tensor, c: Tensor) -> Tensor:
    return 2 / (1 - c * x.pow(2).sum(dim=-1, keepdim=True))


def p_mobius_add(x: Tensor, y: Tensor, c: Tensor) -> Tensor:
    x2 = x.pow(2).sum(dim=-1, keepdim=True)
    y2 = y.pow(2).sum(dim=-1, keepdim=True)
    xy = (x * y).sum(dim=-1, keepdim=True)
    num = (1 + 2 * c * xy + c * y2) * x + (1 - c * x2) * y
    denom = 1 + 2 * c * xy + c**2 * x2 * y2
    return num / denom.clamp_min(P_MIN_NORM)


def p_exp_map_mu0(x: Tensor, R: Tensor) -> Tensor:  # poincare.py:132-137 -> geoopt expmap0
    c = _p_c(R)
    sc = c**0.5
    n = x.norm(dim=-1, p=2, keepdim=True).clamp_min(P_MIN_NORM)
    return _p_tanh(sc * n) * x / (sc * n)


def p_exp_map(u: Tensor, at: Tensor, R: Tensor) -> Tensor:  # poincare.py:124-129 -> geoopt expmap
    c = _p_c(R)
    sc = c**0.5
    n = u.norm(dim=-1, p=2, keepdim=True).clamp_min(P_MIN_NORM)
    second = _p_tanh(sc / 2 * p_lambda_x(at, c) * n) * u / (sc * n)
    return p_mobius_add(at, second, c)


def p_log_map(y: Tensor, at: Tensor, R: Tensor) -> Tensor:  # poincare.py:140-145 -> geoopt logmap
    c = _p_c(R)
    sc = c**0.5
    sub = p_mobius_add(-at, y, c)
    sn = sub.norm(dim=-1, p=2, keepdim=True).clamp_min(P_MIN_NORM)
    return 2 / sc / p_lambda_x(at, c) * _PArtanh.apply(sc * sn) * sub / sn


def p_log_map_mu0(y: Tensor, R: Tensor) -> Tensor:  # poincare.py:148-149 -> geoopt logmap0
    sc = _p_c(R)**0.5
    n = y.norm(dim=-1, p=2, keepdim=True).clamp_min(P_MIN_NORM)
    return y / n / sc * _PArtanh.apply(sc * n)


def p_pt_mu0(x: Tensor, dst: Tensor, R: Tensor) -> Tensor:  # poincare.py:116-117 -> geoopt parallel_transport0
    return x * (1 - _p_c(R) * dst.pow(2).sum(dim=-1, keepdim=True)).clamp_min(P_MIN_NORM)


def p_inv_pt_mu0(x: Tensor, src: Tensor, R: Tensor) -> Tensor:  # poincare.py:120-121 -> geoopt parallel_transport0back
    return x / (1 - _p_c(R) * src.pow(2).sum(dim=-1, keepdim=True)).clamp_min(P_MIN_NORM)


def p_sample_projection_mu0(v: Tensor, at: Tensor, R: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:
    u = v / p_lambda_x(at, _p_c(R))  # poincare.py:152-157
    return p_exp_map(u, at, R), (u, v)


def p_inverse_sample_projection_mu0(z: Tensor, at: Tensor, R: Tensor) -> Tuple[Tensor, Tensor]:
    u = p_log_map(z, at, R)  # poincare.py:160-164
    return u, u * p_lambda_x(at, _p_c(R))


def poincare_to_lorentz(y: Tensor, R: Tensor) -> Tensor:  # poincare.py:167-170 (reference-owned)
    n2 = torch.norm(y, p=2, dim=-1, keepdim=True)**2
    return torch.cat((R * (R**2 + n2), 2 * R**2 * y), dim=-1) / (R**2 - n2)


def p_logdet(mu: Tensor, z: Tensor, R: Tensor) -> Tensor:  # poincare.py:55-89 (reference-owned: via the Lorentz model)
    if z.dim() > mu.dim():
        mu = mu.unsqueeze(0).expand(z.shape)
    u, _ = h_inverse_sample_projection_mu0(poincare_to_lorentz(z, R), poincare_to_lorentz(mu, R), R)
    return h_logdet(u, R)


# ------------------------------------------------------------- stereographically projected sphere `d`
# spherical_projected.py.  Everything here is reference-owned EXCEPT mob_add, which calls geoopt's mobius_add with
# c = -K (spherical_projected.py:113): that one formula is PARITY UNPINNED like the Poincare ball; the functions that
# do not touch it (exp_map_mu0, inverse_exp_map_mu0, the two parallel transports, lambda_x, projected_to_spherical,
# spherical_projected_distance) are pinned by tests/golden/g6_projected.npz.
D_MIN_NORM = 1e-15  # spherical_projected.py:26


def d_lambda_x_c(x: Tensor, c: Tensor) -> Tensor:  # spherical_projected.py:124-125
    return 2 / (1 + c * x.pow(2).sum(dim=-1, keepdim=True)).clamp(min=D_MIN_NORM)


def d_lambda_x(x: Tensor, R: Tensor) -> Tensor:  # spherical_projected.py:128-129
    return d_lambda_x_c(x, 1 / R**2)


def d_mob_add(x: Tensor, y: Tensor, K: Tensor) -> Tensor:  # spherical_projected.py:107-113 -> geoopt mobius_add(c=-K)
    return p_mobius_add(x, y, -K)


def d_pt_mu0(x: Tensor, dst: Tensor, R: Tensor) -> Tensor:  # spherical_projected.py:140-141
    return (2 / d_lambda_x(dst, R)) * x


def d_inv_pt_mu0(x: Tensor, src: Tensor, R: Tensor) -> Tensor:  # spherical_projected.py:144-145
    return (d_lambda_x(src, R) / 2) * x


def d_exp_map(x: Tensor, at: Tensor, R: Tensor) -> Tensor:  # spherical_projected.py:148-154
    r = torch.norm(x, p=2, dim=-1, keepdim=True).clamp(min=D_MIN_NORM) / R
    c = 1 / R**2
    arg = r * d_lambda_x_c(at, c) / 2
    rhs = torch.tan(arg) * x / r
    return d_mob_add(at, rhs, c)


def d_exp_map_mu0(x: Tensor, R: Tensor) -> Tensor:  # spherical_projected.py:157-161
    r = torch.norm(x, p=2, dim=-1, keepdim=True).clamp(min=D_MIN_NORM) / R
    return torch.tan(r) * x / r


def d_log_map(x: Tensor, at: Tensor, R: Tensor) -> Tensor:  # spherical_projected.py:164-169
    c = 1 / R**2
    mxpy = d_mob_add(-at, x, c)
    nmxpy = torch.norm(mxpy, p=2, dim=-1, keepdim=True).clamp(min=D_MIN_NORM) / R
    normalized = mxpy / nmxpy
    return 2 / d_lambda_x_c(at, c) * torch.atan(nmxpy) * normalized


def d_log_map_mu0(x: Tensor, R: Tensor) -> Tensor:  # spherical_projected.py:172-175
    nx = torch.norm(x, p=2, dim=-1, keepdim=True).clamp(min=D_MIN_NORM) / R
    return torch.atan(nx) * (x / nx)


def d_sample_projection_mu0(v: Tensor, at: Tensor, R: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:
    u = v / d_lambda_x(at, R)  # spherical_projected.py:178-181
    return d_exp_map(u, at, R), (u, v)


def d_inverse_sample_projection_mu0(z: Tensor, at: Tensor, R: Tensor) -> Tuple[Tensor, Tensor]:
    u = d_log_map(z, at, R)  # spherical_projected.py:184-188
    return u, u * d_lambda_x_c(at, 1 / R**2)


def projected_to_spherical(y: Tensor, R: Tensor) -> Tensor:  # spherical_projected.py:191-196
    yn2 = torch.norm(y, p=2, dim=-1, keepdim=True)**2
    r2 = R * R
    return torch.cat((R * (r2 - yn2), 2 * r2 * y), dim=-1) / (yn2 + r2)


def d_logdet(mu: Tensor, z: Tensor, R: Tensor) -> Tensor:  # spherical_projected.py:56-88: through the sphere
    if z.dim() > mu.dim():
        mu = mu.unsqueeze(0).expand(z.shape)
    u, _ = s_inverse_sample_projection_mu0(projected_to_spherical(z, R), projected_to_spherical(mu, R), R)
    return s_logdet(u, R)


def spherical_projected_distance(x: Tensor, y: Tensor, K: Tensor) -> Tensor:  # spherical_projected.py:91-98
    diff = x - y
    nd = torch.sum(diff * diff, dim=-1, keepdim=True)
    nx = torch.sum(x * x, dim=-1, keepdim=True)
    ny = torch.sum(y * y, dim=-1, keepdim=True)
    return 1. / sqrt(K) * torch.acos(torch.clamp(1 - 2 * K * nd / ((1 + K * nx) * (1 + K * ny)), max=1.0))


def spherical_projected_gyro_distance(x: Tensor, y: Tensor, K: Tensor) -> Tensor:  # spherical_projected.py:101-105
    sk = sqrt(K)
    return 2. / sk * torch.atan(sk * torch.norm(d_mob_add(-x, y, K), p=2, dim=-1, keepdim=True))


# ------------------------------------------------------------- geodesic distances
# The reference keeps these next to its operators: as helpers of its own op tests (h, s, e) and in the ops modules (p, d).
def h_distance(x: Tensor, y: Tensor, R: Tensor) -> Tensor:  # tests/mvae/ops/test_hyperbolics.py:46-47
    return R * acosh(-lorentz_product(x, y, keepdim=True) / (R**2))


def s_distance(x: Tensor, y: Tensor, R: Tensor) -> Tensor:  # tests/mvae/ops/test_spherical.py:45-48
    ndot = torch.sum(x * y, dim=-1, keepdim=True) / R**2
    return R * torch.acos(torch.clamp(ndot, min=-1., max=1.))


def e_distance(x: Tensor, y: Tensor) -> Tensor:  # tests/mvae/ops/test_euclidean.py:41-42
    return 2 * torch.norm(x - y, dim=-1, p=2, keepdim=True)


def p_distance(x: Tensor, y: Tensor, R: Tensor) -> Tensor:  # poincare.py:92-105 (mobius_add: PARITY UNPINNED)
    c = _p_c(R)
    sqrt_c = sqrt(c)
    mob = p_mobius_add(-x, y, c).norm(dim=-1, p=2, keepdim=True)
    return atanh(sqrt_c * mob) * 2 / sqrt_c


# ------------------------------------------------------------- universal manifold `u`  (universal.py:28-83)
U_EPS = 1e-6  # universal.py:53, component.py:231


def u_radius(K: Tensor) -> Tensor:  # universal.py:30-32
    return torch.relu(1 / sqrt(K.abs()))


def u_choice(K: Tensor, eps: float = U_EPS) -> int:  # universal.py:67-74: -1 Poincare ball, +1 projected sphere, 0 Euclid
    k = float(K.detach())
    return -1 if k < -eps else (1 if k > eps else 0)
