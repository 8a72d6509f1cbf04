"""Tensor-level wrappers over the C ABI (include/mvae_hip.h): torch tensors in, torch tensors out.

torch is used for device memory and the current stream only; all arithmetic happens in libmvae_hip.so.
Inputs must be float32 tensors on a HIP device; there is no CPU path.
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import ComponentDesc, check, load, ptr, stream_ptr

KIND_OF_LETTER = {"e": _lib.EUCLIDEAN, "h": _lib.HYPERBOLOID, "s": _lib.SPHERE, "p": _lib.POINCARE,
                  "d": _lib.PROJ_SPHERE, "u": _lib.UNIVERSAL}
_PROJECTED = (_lib.POINCARE, _lib.PROJ_SPHERE, _lib.UNIVERSAL)  # tangent dim = ambient dim; logdet takes (mu, z)


def ambient_dim(kind: int, d: int) -> int:
    return d + 1 if kind in (_lib.HYPERBOLOID, _lib.SPHERE) else d


def _f32c(t: Tensor) -> Tensor:
    if not t.is_cuda:
        raise _lib.MvaeHipError("mvae_amd ops need tensors on a HIP device (there is no CPU path)")
    if t.dtype != torch.float32:
        raise _lib.MvaeHipError(f"the HIP path computes in float32, got {t.dtype} (use --doubles=False)")
    return t.contiguous()


def colsum(G: Tensor) -> Tensor:
    """out[N] = column sums of G[M, N], fixed summation order (mvae_colsum)."""
    G = _f32c(G)
    M, N = G.shape
    out = G.new_empty(N)
    nws = int(load().mvae_colsum_workspace_floats(M, N))
    ws = G.new_empty(max(nws, 1)) if nws > 0 else None
    check(load().mvae_colsum(ptr(G), ptr(out), M, N, ptr(ws), stream_ptr(G.device)))
    return out


# --------------------------------------------------------------------------------------------- primitives
# Every manifold primitive is one forward kernel (mvae_<name>) and one backward kernel (mvae_primitive_backward, which
# evaluates the same device template over dual numbers, so the reference's custom derivative rules -- LeakyClamp, Acosh,
# hard clamps, norm'(0) = 0 -- are the ones torch.autograd sees).  `_Prim` is the torch.autograd.Function joining them;
# it is only entered when an input requires grad.
_FWD = {
    _lib.OP_EXP0: lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_exp_map_mu0(k, a, o1, rows, d, r, st),
    _lib.OP_LOG0: lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_inverse_exp_map_mu0(k, a, o1, rows, d, r, st),
    _lib.OP_PT0: lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_parallel_transport_mu0(k, a, b, o1, rows, d, r,
                                                                                                  st),
    _lib.OP_IPT0: lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_inverse_parallel_transport_mu0(
        k, a, b, o1, rows, d, r, st),
    _lib.OP_SAMPLE: lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_sample_projection_mu0(
        k, a, b, o1, o2, rows, atr, d, r, st),
    _lib.OP_ISAMPLE: lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_inverse_sample_projection_mu0(
        k, a, b, o1, o2, rows, atr, d, r, st),
    _lib.OP_LOGDET: lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_logdet(k, a, b, c, o1, rows, atr, d, r, st),
    _lib.OP_EXP: lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_exp_map(k, a, b, o1, rows, atr, d, r, st),
    _lib.OP_LOG: lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_inverse_exp_map(k, a, b, o1, rows, atr, d, r,
                                                                                           st),
    _lib.OP_DIST: lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_geodesic_distance(
        k, _lib.DIST_GEODESIC, a, b, o1, rows, atr, d, r, st),
    _lib.OP_DIST_GYRO: lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_geodesic_distance(
        k, _lib.DIST_GYRO, a, b, o1, rows, atr, d, r, st),
}
for _op in (_lib.OP_NORMAL_LOGPROB, _lib.OP_NORMAL_RSAMPLE, _lib.OP_NORMAL_KL):
    _FWD[_op] = (lambda op_: (lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_normal_op(
        op_, a, b, c, o1, rows, atr, d, st)))(_op)
for _op in (_lib.OP_LPROD, _lib.OP_LNORM, _lib.OP_TO_BALL, _lib.OP_TO_AMBIENT, _lib.OP_LAMBDA, _lib.OP_MOBADD):
    _FWD[_op] = (lambda op_: (lambda L, k, a, b, c, o1, o2, rows, atr, d, r, st: L.mvae_manifold_aux(
        op_, k, a, b, o1, rows, d, r, st)))(_op)


_NORMAL_OPS = (_lib.OP_NORMAL_LOGPROB, _lib.OP_NORMAL_RSAMPLE, _lib.OP_NORMAL_KL)


def _prim_shape(op: int, kind: int, d: int) -> Tuple[int, int, int, int, int]:
    """(na, nb, nc, n1, n2): host mirror of prim_shape() in csrc/mvae_api.hip."""
    A = ambient_dim(kind, d)
    proj = kind in _PROJECTED
    if op == _lib.OP_EXP0:
        return d, 0, 0, A, 0
    if op == _lib.OP_LOG0:
        return A, 0, 0, A, 0
    if op in (_lib.OP_PT0, _lib.OP_IPT0, _lib.OP_EXP, _lib.OP_LOG, _lib.OP_MOBADD):
        return A, A, 0, A, 0
    if op in (_lib.OP_LNORM, _lib.OP_LAMBDA):
        return A, 0, 0, 1, 0
    if op == _lib.OP_TO_BALL:
        return A, 0, 0, d, 0
    if op == _lib.OP_TO_AMBIENT:
        return d, 0, 0, d + 1, 0
    if op == _lib.OP_NORMAL_LOGPROB:
        return d, d, d, 1, 0
    if op == _lib.OP_NORMAL_RSAMPLE:
        return d, d, d, d, 0
    if op == _lib.OP_NORMAL_KL:
        return d, 0, d, 1, 0
    if op == _lib.OP_SAMPLE:
        return d, A, 0, A, A
    if op == _lib.OP_ISAMPLE:
        return A, A, 0, A, d
    if op == _lib.OP_LOGDET:
        return (0, A, A, 1, 0) if proj else ((0, 0, 0, 1, 0) if kind == _lib.EUCLIDEAN else (A, 0, 0, 1, 0))
    return A, A, 0, 1, 0


def _launch_fwd(op, kind, a, b, c3, rp, rows, at_rows, d, n1, n2, like):
    o1 = like.new_empty(rows, n1)
    o2 = like.new_empty(rows, n2) if n2 else None
    check(_FWD[op](load(), kind, ptr(a), ptr(b), ptr(c3), ptr(o1), ptr(o2), rows, at_rows, d, ptr(rp),
                   stream_ptr(like.device)))
    return o1, o2


class _Prim(torch.autograd.Function):
    """inputs a[rows, na], b[at_rows, nb], c3[rows, nc] (2-D, contiguous, or None) and the radius / curvature
    parameter rp[1] (or None) -> o1[rows, n1], o2[rows, n2] (or None)."""

    @staticmethod
    def forward(ctx, a, b, c3, rp, op, kind, d, rows, at_rows):
        na, nb, nc, n1, n2 = _prim_shape(op, kind, d)
        like = a if a is not None else b
        o1, o2 = _launch_fwd(op, kind, a, b, c3, rp, rows, at_rows, d, n1, n2, like)
        ctx.save_for_backward(a, b, c3, rp)
        ctx.meta = (op, kind, d, rows, at_rows)
        if o2 is None:
            o2 = like.new_empty(0)
            ctx.mark_non_differentiable(o2)
        return o1, o2

    @staticmethod
    def backward(ctx, g1, g2):
        a, b, c3, rp = ctx.saved_tensors
        op, kind, d, rows, at_rows = ctx.meta
        na, nb, nc, n1, n2 = _prim_shape(op, kind, d)
        like = a if a is not None else b
        g1 = like.new_zeros(rows, n1) if g1 is None else _f32c(g1)
        g2 = None if (n2 == 0 or g2 is None) else _f32c(g2)
        need = ctx.needs_input_grad
        ga = like.new_zeros(rows, na) if (need[0] and na) else None
        gb = like.new_zeros(rows, nb) if (need[1] and nb) else None
        gc = like.new_zeros(rows, nc) if (need[2] and nc) else None
        gr = like.new_zeros(rows, 1) if (need[3] and rp is not None) else None
        check(load().mvae_primitive_backward(op, kind, ptr(a), ptr(b), ptr(c3), ptr(g1), ptr(g2), ptr(ga), ptr(gb),
                                             ptr(gc), ptr(gr), rows, at_rows, d, ptr(rp), stream_ptr(like.device)))
        if gb is not None and at_rows < rows:  # b was broadcast over leading sample dims: sum them (index order)
            gb = colsum(gb.view(rows // at_rows, at_rows * nb)).view(at_rows, nb)
        if gc is not None and at_rows < rows and op in _NORMAL_OPS:  # the normal ops broadcast c3 as well
            gc = colsum(gc.view(rows // at_rows, at_rows * nc)).view(at_rows, nc)
        if gr is not None:
            gr = colsum(gr)
        return ga, gb, gc, gr, None, None, None, None, None


def _radius_param(kind: int, radius, like: Tensor) -> Optional[Tensor]:
    """The radius (or, for `u`, curvature) as a 1-element float32 device tensor that keeps its autograd history."""
    if kind == _lib.EUCLIDEAN:
        return None
    if radius is None:
        raise _lib.MvaeHipError("this manifold needs a radius")
    if not torch.is_tensor(radius):
        radius = torch.tensor(float(radius), dtype=torch.float32, device=like.device)
    if radius.dtype != torch.float32:
        raise _lib.MvaeHipError(f"the HIP path computes in float32, got a {radius.dtype} radius")
    if radius.numel() != 1:
        raise _lib.MvaeHipError("one radius per call: per-row radii are not supported")
    return radius.to(like.device).reshape(1).contiguous()


def _rows2d(t: Optional[Tensor], n: int) -> Optional[Tensor]:
    return None if t is None else _f32c(t).reshape(-1, n)


def _prim(op: int, kind: int, d: int, lead, a, b, c3, radius):
    """Runs primitive `op`.  a / c3 have leading dims `lead`; b's leading dims are `lead` or a suffix of them
    (broadcast over the missing sample dims).  Returns (o1, o2) shaped lead + (n,)."""
    na, nb, nc, n1, n2 = _prim_shape(op, kind, d)
    like = a if a is not None else c3
    rows = 1
    for s_ in lead:
        rows *= int(s_)
    a2 = _rows2d(a, na) if na else None
    b2, at_rows = None, rows

    def bcast(t, n):  # leading dims `lead`, or a suffix of them (broadcast over the missing sample dims)
        t = _f32c(t)
        tl = tuple(t.shape[:-1])
        if tl != tuple(lead) and not (len(tl) <= len(lead) and tuple(lead[len(lead) - len(tl):]) == tl):
            t = t.expand(tuple(lead) + (n,)).contiguous()
        return t.reshape(-1, n)

    if nb:
        b2 = bcast(b, nb)
        at_rows = max(1, b2.shape[0])
    if nc and op in _NORMAL_OPS:
        if nb:
            c3 = torch.broadcast_to(c3, b.shape[:-1] + (nc,)) if tuple(c3.shape[:-1]) != tuple(b.shape[:-1]) else c3
        c2 = bcast(c3, nc)
        if nb and c2.shape[0] != at_rows:
            raise ValueError("loc and scale must have the same leading dims")
        if not nb:
            at_rows = max(1, c2.shape[0])
    else:
        c2 = _rows2d(c3, nc) if nc else None
    rp = _radius_param(kind, radius, like)
    if rows == 0:
        return like.new_empty(tuple(lead) + (n1,)), (like.new_empty(tuple(lead) + (n2,)) if n2 else None)
    tensors = (a2, b2, c2, rp)
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        o1, o2 = _Prim.apply(a2, b2, c2, rp, op, kind, d, rows, at_rows)
    else:
        det = [None if t is None else t.detach() for t in tensors]
        o1, o2 = _launch_fwd(op, kind, det[0], det[1], det[2], det[3], rows, at_rows, d, n1, n2,
                             det[0] if det[0] is not None else det[1])
    o1 = o1.view(tuple(lead) + (n1,))
    o2 = o2.view(tuple(lead) + (n2,)) if n2 else None
    return o1, o2


def _true_dim(kind: int, ambient: int) -> int:
    return ambient - 1 if kind in (_lib.HYPERBOLOID, _lib.SPHERE) else ambient


def exp_map_mu0(kind: int, x: Tensor, radius=None) -> Tensor:
    return _prim(_lib.OP_EXP0, kind, x.shape[-1], x.shape[:-1], x, None, None, radius)[0]


def inverse_exp_map_mu0(kind: int, x: Tensor, radius=None) -> Tensor:
    return _prim(_lib.OP_LOG0, kind, _true_dim(kind, x.shape[-1]), x.shape[:-1], x, None, None, radius)[0]


def _two(op: int, kind: int, x: Tensor, other: Tensor, radius) -> Tensor:
    x, other = torch.broadcast_tensors(x, other)
    return _prim(op, kind, _true_dim(kind, x.shape[-1]), x.shape[:-1], x, other, None, radius)[0]


def parallel_transport_mu0(kind: int, x: Tensor, dst: Tensor, radius=None) -> Tensor:
    return _two(_lib.OP_PT0, kind, x, dst, radius)


def inverse_parallel_transport_mu0(kind: int, x: Tensor, src: Tensor, radius=None) -> Tensor:
    return _two(_lib.OP_IPT0, kind, x, src, radius)


def exp_map(kind: int, x: Tensor, at_point: Tensor, radius=None) -> Tensor:
    """Manifold exp map of the tangent vector x at `at_point` (hyperbolics.py:106-111 and its siblings)."""
    return _prim(_lib.OP_EXP, kind, _true_dim(kind, x.shape[-1]), x.shape[:-1], x, at_point, None, radius)[0]


def inverse_exp_map(kind: int, x: Tensor, at_point: Tensor, radius=None) -> Tensor:
    return _prim(_lib.OP_LOG, kind, _true_dim(kind, x.shape[-1]), x.shape[:-1], x, at_point, None, radius)[0]


def geodesic_distance(kind: int, x: Tensor, y: Tensor, radius=None, gyro: bool = False, keepdim: bool = True) -> Tensor:
    """Geodesic distance of the manifold (include/mvae_hip.h: mvae_geodesic_distance), [..., 1] like the reference's
    helpers (keepdim=True)."""
    x, y = torch.broadcast_tensors(x, y)
    out = _prim(_lib.OP_DIST_GYRO if gyro else _lib.OP_DIST, kind, _true_dim(kind, x.shape[-1]), x.shape[:-1], x, y,
                None, radius)[0]
    return out if keepdim else out.squeeze(-1)


def sample_projection_mu0(kind: int, v: Tensor, at_point: Tensor, radius=None):
    z, u = _prim(_lib.OP_SAMPLE, kind, v.shape[-1], v.shape[:-1], v, at_point, None, radius)
    return z, (u, v)


def inverse_sample_projection_mu0(kind: int, z: Tensor, at_point: Tensor, radius=None):
    u, v = _prim(_lib.OP_ISAMPLE, kind, _true_dim(kind, z.shape[-1]), z.shape[:-1], z, at_point, None, radius)
    return u, v


def logdet(kind: int, u: Optional[Tensor], mu: Optional[Tensor], z: Optional[Tensor], radius=None) -> Tensor:
    if kind in _PROJECTED:
        out = _prim(_lib.OP_LOGDET, kind, z.shape[-1], z.shape[:-1], None, mu, z, radius)[0]
    else:
        out = _prim(_lib.OP_LOGDET, kind, _true_dim(kind, u.shape[-1]), u.shape[:-1], u, None, None, radius)[0]
    return out.squeeze(-1)


def normal_log_prob(value: Tensor, loc: Tensor, scale: Tensor) -> Tensor:
    """sum_i log N(value_i; loc_i, scale_i): EuclideanNormal.log_prob (wrapped_distributions.py:39-42); loc / scale may
    lack leading sample dims of `value`."""
    loc, scale = torch.broadcast_tensors(loc, scale)
    return _prim(_lib.OP_NORMAL_LOGPROB, _lib.EUCLIDEAN, value.shape[-1], value.shape[:-1], value, loc, scale,
                 None)[0].squeeze(-1)


def normal_rsample(eps: Tensor, loc: Tensor, scale: Tensor) -> Tensor:
    """loc + eps * scale (torch.distributions.Normal.rsample with the standard-normal draw given)."""
    loc, scale = torch.broadcast_tensors(loc, scale)
    return _prim(_lib.OP_NORMAL_RSAMPLE, _lib.EUCLIDEAN, eps.shape[-1], eps.shape[:-1], eps, loc, scale, None)[0]


def normal_kl_standard(loc: Tensor, scale: Tensor) -> Tensor:
    """KL(N(loc, scale) || N(0, 1)) summed over the last dim (sampling_procedures.py:153-155)."""
    loc, scale = torch.broadcast_tensors(loc, scale)
    return _prim(_lib.OP_NORMAL_KL, _lib.EUCLIDEAN, loc.shape[-1], loc.shape[:-1], loc, None, scale, None)[0].squeeze(-1)


def manifold_aux(op: int, kind: int, x: Tensor, y: Optional[Tensor] = None, radius=None) -> Tensor:
    """The small public helpers of the reference's ops modules (mvae_manifold_aux): Lorentz product / norm, the model
    conversions, the conformal factor and Moebius addition.  Output [..., n]."""
    if y is not None:
        x, y = torch.broadcast_tensors(x, y)
    n_in = x.shape[-1]
    d = n_in if op == _lib.OP_TO_AMBIENT else _true_dim(kind, n_in)
    return _prim(op, kind, d, x.shape[:-1], x, y, None, radius)[0]


class _ScalarFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, name, lo, hi):
        y, dy = scalar_fn(name, x.detach(), lo, hi)
        ctx.save_for_backward(dy)
        return y

    @staticmethod
    def backward(ctx, g):
        (dy,) = ctx.saved_tensors
        return elementwise_mul(_f32c(g), dy), None, None, None


def guarded(name: str, x: Tensor, lo: float = float("-inf"), hi: float = float("inf")) -> Tensor:
    """Differentiable form of scalar_fn: f(x) with the reference's custom backward rule (ops/common.py:28-147)."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _ScalarFn.apply(x, name, lo, hi)
    return scalar_fn(name, x.detach(), lo, hi)[0]


def elementwise_mul(a: Tensor, b: Tensor) -> Tensor:
    """a * b on the device (mvae_scalar_fn's companion: the chain-rule product of _ScalarFn.backward)."""
    a, b = _f32c(a), _f32c(b)
    out = torch.empty_like(a)
    check(load().mvae_mul(ptr(a), ptr(b), ptr(out), a.numel(), stream_ptr(a.device)))
    return out


def scalar_fn(name: str, x: Tensor, lo: float = float("-inf"), hi: float = float("inf")):
    """(f(x), f'(x)) of one of the reference's guarded scalar functions (ops/common.py:28-147) through the device code
    of the manifold kernels (mvae_scalar_fn)."""
    x = _f32c(x)
    y, dy = torch.empty_like(x), torch.empty_like(x)
    check(load().mvae_scalar_fn(_lib.SCALAR_FNS[name], ptr(x), ptr(y), ptr(dy), x.numel(), float(lo), float(hi),
                                stream_ptr(x.device)))
    return y, dy


# --------------------------------------------------------------------------------------------- components
class ComponentLayout:
    """Column layout of the fused head matrix / eps / concat_z for a list of (letter, true_dim)."""

    def __init__(self, comps: Sequence[Tuple[str, int]], scalar_parametrization: bool = False):
        self.comps = list(comps)
        n = len(self.comps)
        if n < 1 or n > _lib.MAX_COMPONENTS:
            raise ValueError(f"between 1 and {_lib.MAX_COMPONENTS} components are supported, got {n}")
        self.descs = (ComponentDesc * n)()
        mean_col = 0
        total_true = sum(d for _, d in self.comps)
        logvar_col = total_true
        eps_col = z_col = 0
        for i, (letter, d) in enumerate(self.comps):
            kind = KIND_OF_LETTER[letter]
            if d > _lib.MAX_TRUE_DIM:
                raise NotImplementedError(f"true_dim {d} > {_lib.MAX_TRUE_DIM}")
            lvd = 1 if scalar_parametrization else d
            self.descs[i] = ComponentDesc(kind, d, mean_col, logvar_col, lvd, eps_col, z_col, i)
            mean_col += d
            logvar_col += lvd
            eps_col += d
            z_col += ambient_dim(kind, d)
        self.heads_dim = logvar_col
        self.eps_dim = eps_col
        self.z_dim = z_col
        self.n = n


# ---- float64 latent chain (the reference CLI's default numerics, run.py:77,98-101): while the switch is on, the component
# operators evaluate softplus / exp map / transport / log map / log-det / log-probabilities and their derivatives in double
# between float32 tensors (mvae_component_forward_f64 / _backward_f64).  Dense layers stay float32.
_FLOAT64_CHAIN = False


def set_float64_chain(on: bool) -> bool:
    """Switch the component operators to the float64 chain (process-wide); returns the previous setting."""
    global _FLOAT64_CHAIN
    prev, _FLOAT64_CHAIN = _FLOAT64_CHAIN, bool(on)
    return prev


class float64_chain:
    """with float64_chain(): ... -- the component operators inside run their chain in float64."""

    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        self.prev = set_float64_chain(self.on)

    def __exit__(self, *exc):
        set_float64_chain(self.prev)
        return False


def component_forward(layout: ComponentLayout, heads: Tensor, eps: Tensor, radii: Optional[Tensor],
                      want_kl: bool = True, want_log_probs: bool = False, want_params: bool = False):
    """heads [B, heads_dim]; eps [B, eps_dim] or [n, B, eps_dim].  Returns dict(z, kl, log_q, log_p, mu, std)."""
    heads, eps = _f32c(heads), _f32c(eps)
    head_rows = heads.shape[0]
    rows = eps.numel() // layout.eps_dim
    lead = eps.shape[:-1]
    z = eps.new_empty(lead + (layout.z_dim,))
    kl = eps.new_empty((layout.n,) + lead) if want_kl else None
    lq = eps.new_empty((layout.n,) + lead) if want_log_probs else None
    lp = eps.new_empty((layout.n,) + lead) if want_log_probs else None
    mu = heads.new_empty(head_rows, layout.z_dim) if want_params else None
    sd = heads.new_zeros(head_rows, layout.eps_dim) if want_params else None
    radii = None if radii is None else _f32c(radii)
    fn = load().mvae_component_forward_f64 if _FLOAT64_CHAIN else load().mvae_component_forward
    check(fn(layout.descs, layout.n, ptr(heads), heads.shape[-1], ptr(eps), layout.eps_dim, ptr(radii), ptr(z), layout.z_dim,
             ptr(kl), ptr(lq), ptr(lp), ptr(mu), ptr(sd), rows, head_rows, stream_ptr(heads.device)))
    return {"z": z, "kl": kl, "log_q": lq, "log_p": lp, "mu": mu, "std": sd}


def component_backward(layout: ComponentLayout, heads: Tensor, eps: Tensor, radii: Optional[Tensor], dz: Tensor,
                       dkl: Optional[Tensor], dkl_scalar: float = 0.0, want_dradii: bool = True,
                       out_dheads: Optional[Tensor] = None, workspace: Optional[Tensor] = None,
                       out_dradii: Optional[Tensor] = None):
    """dheads[rows, heads_dim] and dradii[ncomp] (a fixed-order sum over the rows: bit-reproducible).  out_dradii: a
    contiguous [ncomp] destination (e.g. the radii region of a flat gradient buffer), every entry is written."""
    heads, eps, dz = _f32c(heads), _f32c(eps), _f32c(dz)
    rows = heads.shape[0]
    dheads = torch.zeros_like(heads) if out_dheads is None else out_dheads
    dradii = (heads.new_zeros(layout.n) if out_dradii is None else out_dradii) if want_dradii else None
    if want_dradii and workspace is None:
        workspace = heads.new_empty(max(1, layout.n * rows))
    dkl = None if dkl is None else _f32c(dkl)
    radii = None if radii is None else _f32c(radii)
    fn = load().mvae_component_backward_f64 if _FLOAT64_CHAIN else load().mvae_component_backward
    check(fn(layout.descs, layout.n, ptr(heads), heads.shape[-1], ptr(eps), layout.eps_dim, ptr(radii), ptr(dz), layout.z_dim,
             ptr(dkl), float(dkl_scalar), ptr(dheads), ptr(dradii), ptr(workspace if want_dradii else None), rows,
             stream_ptr(heads.device)))
    return dheads, dradii


class _ComponentFn(torch.autograd.Function):
    """(heads[B, NH], radii[ncomp], eps[B, E]) -> (z[B, Z], kl[ncomp, B]) with the fused component operators
    (training path: rows == head_rows)."""

    @staticmethod
    def forward(ctx, heads, radii, eps, layout):
        out = component_forward(layout, heads.detach(), eps, radii.detach(), want_kl=True)
        ctx.save_for_backward(heads.detach(), radii.detach(), eps)
        ctx.layout = layout
        ctx.f64 = _FLOAT64_CHAIN  # the backward pass differentiates the chain the forward pass ran
        return out["z"], out["kl"]

    @staticmethod
    def backward(ctx, dz, dkl):
        heads, radii, eps = ctx.saved_tensors
        lay = ctx.layout
        dz = torch.zeros(heads.shape[0], lay.z_dim, device=heads.device) if dz is None else dz
        dkl = torch.zeros(lay.n, heads.shape[0], device=heads.device) if dkl is None else dkl
        with float64_chain(ctx.f64):
            dheads, dradii = component_backward(lay, heads, eps, radii, dz, dkl, want_dradii=ctx.needs_input_grad[1])
        return (dheads if ctx.needs_input_grad[0] else None), dradii, None, None


def component_rsample_kl(layout: ComponentLayout, heads: Tensor, radii: Tensor, eps: Tensor):
    """Differentiable (z, kl) of all components of `layout` (Component.forward -> rsample_with_parts -> kl_loss)."""
    if torch.is_grad_enabled() and (heads.requires_grad or radii.requires_grad):
        return _ComponentFn.apply(_f32c(heads), _f32c(radii), _f32c(eps), layout)
    out = component_forward(layout, heads.detach(), eps, radii.detach(), want_kl=True)
    return out["z"], out["kl"]


# --------------------------------------------------------------------------------------------- dense layers
def linear_forward(x: Tensor, W: Tensor, b: Optional[Tensor], relu: bool = False) -> Tensor:
    x, W = _f32c(x), _f32c(W)
    K = x.shape[-1]
    N = W.shape[0]
    M = x.numel() // K
    y = x.new_empty(x.shape[:-1] + (N,))
    check(load().mvae_linear_forward(ptr(x), ptr(W), ptr(None if b is None else _f32c(b)), ptr(y), M, N, K,
                                     1 if relu else 0, stream_ptr(x.device)))
    return y


def linear_backward(x: Tensor, W: Tensor, dy: Tensor, relu_in: bool = False, need_dx: bool = True,
                    out_dW: Optional[Tensor] = None, out_db: Optional[Tensor] = None):
    """out_dW / out_db: contiguous destinations (e.g. views into a flat gradient buffer) written in place."""
    x, W, dy = _f32c(x), _f32c(W), _f32c(dy)
    K = x.shape[-1]
    N = W.shape[0]
    M = x.numel() // K
    dW = torch.empty_like(W) if out_dW is None else out_dW
    db = W.new_empty(N) if out_db is None else out_db
    assert dW.is_contiguous() and dW.numel() == N * K and db.is_contiguous() and db.numel() == N
    dx = torch.empty_like(x) if need_dx else None
    check(load().mvae_linear_backward(ptr(x), ptr(W), ptr(dy), 1 if relu_in else 0, ptr(dW), ptr(db), ptr(dx), M, N, K,
                                      stream_ptr(x.device)))
    return dW, db, dx


def relu_mask_(dy: Tensor, y: Tensor) -> Tensor:
    """dy[i] = 0 where y[i] <= 0, in place (backward through a ReLU whose output is y)."""
    check(load().mvae_relu_mask(ptr(dy), ptr(y), dy.numel(), stream_ptr(dy.device)))
    return dy


class _LinearFn(torch.autograd.Function):
    """torch.nn.Linear (+ optional ReLU) on the MFMA kernels: y = act(x W^T + b)."""

    @staticmethod
    def forward(ctx, x, W, b, relu):
        y = linear_forward(x.detach(), W.detach(), None if b is None else b.detach(), relu)
        ctx.save_for_backward(x.detach(), W.detach(), y if relu else None)
        ctx.relu, ctx.has_bias = relu, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        dy = _f32c(dy)
        if ctx.relu:
            dy = relu_mask_(dy.clone(), y)
        x2, dy2 = x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1])
        dW, db, dx = linear_backward(x2, W, dy2, need_dx=ctx.needs_input_grad[0])
        return (None if dx is None else dx.view(x.shape)), dW, (db if ctx.has_bias else None), None


def linear(x: Tensor, W: Tensor, b: Optional[Tensor], relu: bool = False) -> Tensor:
    """Differentiable torch.nn.functional.linear (+ ReLU) on the HIP kernels."""
    if torch.is_grad_enabled() and (x.requires_grad or W.requires_grad or (b is not None and b.requires_grad)):
        return _LinearFn.apply(x, W, b, relu)
    return linear_forward(x.detach(), W.detach(), None if b is None else b.detach(), relu)


# --------------------------------------------------------------------------------------------- log-likelihood pieces
def bce_rows(logits: Tensor, x: Tensor) -> Tensor:
    """sum_j BCE-with-logits(logits[..., j], x[..., j]) with x broadcast over leading sample dims of logits."""
    logits, x = _f32c(logits), _f32c(x)
    D = logits.shape[-1]
    rows, x_rows = logits.numel() // D, x.numel() // D
    out = logits.new_empty(logits.shape[:-1])
    check(load().mvae_bce_rows(ptr(logits), ptr(x), ptr(out), rows, x_rows, D, stream_ptr(logits.device)))
    return out


def decode_bce_rows(z: Tensor, w_d0: Tensor, b_d0: Tensor, w_l: Tensor, b_l: Tensor, x: Tensor) -> Optional[Tensor]:
    """bce_rows(linear(relu(linear(z, w_d0, b_d0)), w_l, b_l), x) in one launch (mvae_decode_bce_rows: the hidden layer and the
    logits never reach memory).  z [..., Z], x [x_rows, D] broadcast over z's leading sample dims.  None if the shape is outside
    what that kernel covers -- the caller then composes the three operators."""
    z, x = _f32c(z), _f32c(x)
    w_d0, b_d0, w_l, b_l = _f32c(w_d0.detach()), _f32c(b_d0.detach()), _f32c(w_l.detach()), _f32c(b_l.detach())
    Z, H, D = z.shape[-1], w_d0.shape[0], w_l.shape[0]
    rows, x_rows = z.numel() // Z, x.numel() // D
    if x.numel() * 4 >= 2 ** 32:  # the kernel addresses the targets with 32-bit byte offsets
        return None
    out = z.new_empty(z.shape[:-1])
    rc = load().mvae_decode_bce_rows(ptr(z), rows, Z, ptr(w_d0), ptr(b_d0), ptr(w_l), ptr(b_l), ptr(x), x_rows, H, D, ptr(out),
                                     stream_ptr(z.device))
    if rc == -2:  # MVAE_E_UNSUPPORTED
        return None
    check(rc)
    return out


def scale_rows(g: Tensor, sc: Tensor) -> Tensor:
    g, sc = _f32c(g), _f32c(sc)
    out = torch.empty_like(g)
    D = g.shape[-1]
    check(load().mvae_scale_rows(ptr(g), ptr(sc), ptr(out), g.numel() // D, D, stream_ptr(g.device)))
    return out


class _BceFn(torch.autograd.Function):
    """sum_j BCE-with-logits per row and its gradient w.r.t. the logits (sigmoid(logits) - x) * upstream[row]."""

    @staticmethod
    def forward(ctx, logits, x):
        logits, x = _f32c(logits.detach()), _f32c(x)
        D = logits.shape[-1]
        bce = logits.new_empty(logits.shape[:-1])
        g = torch.empty_like(logits)
        check(load().mvae_bce_forward_backward(ptr(logits), ptr(x), ptr(bce), ptr(g), logits.numel() // D, D,
                                               stream_ptr(logits.device)))
        ctx.save_for_backward(g)
        return bce

    @staticmethod
    def backward(ctx, dbce):
        (g,) = ctx.saved_tensors
        return scale_rows(g, dbce), None


def bce_with_logits_rows(logits: Tensor, x: Tensor) -> Tensor:
    """Differentiable reconstruction loss summed over the pixels (image_reconstruction.py:81-82, vae.py:131); x has the
    shape of logits (training path) or is broadcast over leading sample dims (evaluation, no grad)."""
    if torch.is_grad_enabled() and logits.requires_grad:
        if x.shape != logits.shape:  # targets broadcast over leading sample dims: the gradient must still flow
            x = x.expand(logits.shape).contiguous()
        return _BceFn.apply(logits, x)
    return bce_rows(logits.detach(), x)


_COV_WS = {}


def loglik_tail(bce: Tensor, log_p: Tensor, log_q: Tensor, z: Tensor, x: Tensor):
    """The tail of ModelVAE.log_likelihood (vae.py:110-121) in two launches: (log p(x) [B], mi [B], cov_norm []) from
    bce [n, B], the per-component log_p / log_q [ncomp, n, B] (component_forward's outputs, summed inside), the samples
    z [n, B, Z] and the targets x [B, D].  None if the shape is outside the covariance kernel's coverage."""
    bce, log_p, log_q, z, x = _f32c(bce), _f32c(log_p), _f32c(log_q), _f32c(z), _f32c(x)
    n, B = bce.shape
    Z, D = z.shape[-1], x.shape[-1]
    if Z > 64 or B * (16 + Z) * 4 > 48 * 1024 or x.shape[0] != B:
        return None
    lib = load()
    st = stream_ptr(bce.device)
    # one workspace (partial sums + arrival counter) per (device, STREAM, D): two calls in flight on different streams must
    # not share a counter.  Zeroed once -- the launch re-arms its counter; a first call inside a graph capture would put the
    # allocation and the fill into the capture, so it is refused there
    key = (bce.device, int(st or 0), D)
    ws = _COV_WS.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        ws = _COV_WS[key] = torch.zeros(int(lib.mvae_cov_norm_workspace_floats(D)), device=bce.device)
    log_px, mi, zmean, cn = bce.new_empty(B), bce.new_empty(B), bce.new_empty(B, Z), bce.new_empty(())
    check(lib.mvae_loglik_reduce_comps(ptr(bce), ptr(log_p), ptr(log_q), log_p.shape[0], ptr(z), Z, ptr(log_px), ptr(mi),
                                       ptr(zmean), n, B, st))
    check(lib.mvae_cov_norm(ptr(x), ptr(zmean), B, D, Z, ptr(ws), ptr(cn), st))
    return log_px, mi, cn


def loglik_reduce(bce: Tensor, log_p: Tensor, log_q: Tensor):
    bce, log_p, log_q = _f32c(bce), _f32c(log_p), _f32c(log_q)
    n, B = bce.shape
    log_px, mi = bce.new_empty(B), bce.new_empty(B)
    check(load().mvae_loglik_reduce(ptr(bce), ptr(log_p), ptr(log_q), ptr(log_px), ptr(mi), n, B,
                                    stream_ptr(bce.device)))
    return log_px, mi


def randn_nonzero(shape, device, generator: Optional[torch.Generator] = None) -> Tensor:
    """N(0, 1) draws of `shape` on the HIP device in ONE launch (mvae_randn), none of them exactly 0 -- what `torch.randn`
    + a zero nudge did in three.  The Philox (seed, offset) pair comes from `generator` (default: the device's torch
    generator) and the generator is advanced, so `torch.manual_seed` / `Generator.manual_seed` reproduce the draws like
    they do torch's own; under stream capture (where a generator's host-side offset cannot be read) torch.randn + the
    nudge are used instead."""
    device = torch.device(device)
    if torch.cuda.is_current_stream_capturing():
        eps = torch.randn(*shape, device=device, generator=generator)
        return eps.masked_fill_(eps == 0, 1e-10)
    gen = generator if generator is not None else torch.cuda.default_generators[
        device.index if device.index is not None else torch.cuda.current_device()]
    seed, offset = int(gen.initial_seed()), int(gen.get_offset())
    out = torch.empty(*shape, device=device, dtype=torch.float32)
    gen.set_offset(offset + 4)  # (torch wants multiples of 4; one tick per call: the item index is part of the counter)
    with torch.cuda.device(device):
        check(load().mvae_randn(ptr(out), out.numel(), seed & 0xFFFFFFFFFFFFFFFF, offset, stream_ptr(device)))
    return out
