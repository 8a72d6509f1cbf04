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


def _radius_arg(kind: int, radius: Optional[Tensor], like: Tensor) -> Optional[Tensor]:
    if kind == _lib.EUCLIDEAN:
        return None
    if radius is None:
        raise _lib.MvaeHipError("this manifold needs a radius")
    if not torch.is_tensor(radius):
        radius = torch.tensor(float(radius), dtype=torch.float32, device=like.device)
    return _f32c(radius.detach().reshape(1).to(like.device))


def _no_grad_inputs(*ts) -> None:
    """The radius is exempt: it is the live nn.Parameter by construction (RadiusManifold takes a callable returning
    it), while a tensor argument that requires grad means the caller expects autograd to flow through."""
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in ts):
        raise NotImplementedError(
            "the standalone manifold primitives are forward-only; differentiable paths go through the fused "
            "component / step operators (Component.forward, ModelVAE.train_step)")


# --------------------------------------------------------------------------------------------- primitives
def exp_map_mu0(kind: int, x: Tensor, radius: Optional[Tensor] = None) -> Tensor:
    _no_grad_inputs(x)
    x = _f32c(x)
    d = x.shape[-1]
    out = x.new_empty(x.shape[:-1] + (ambient_dim(kind, d),))
    r = _radius_arg(kind, radius, x)
    check(load().mvae_exp_map_mu0(kind, ptr(x), ptr(out), x.numel() // d, d, ptr(r), stream_ptr(x.device)))
    return out


def _true_dim(kind: int, ambient: int) -> int:
    return ambient - 1 if kind in (_lib.HYPERBOLOID, _lib.SPHERE) else ambient


def inverse_exp_map_mu0(kind: int, x: Tensor, radius: Optional[Tensor] = None) -> Tensor:
    _no_grad_inputs(x)
    x = _f32c(x)
    A = x.shape[-1]
    out = torch.empty_like(x)
    r = _radius_arg(kind, radius, x)
    check(load().mvae_inverse_exp_map_mu0(kind, ptr(x), ptr(out), x.numel() // A, _true_dim(kind, A), ptr(r),
                                          stream_ptr(x.device)))
    return out


def _pt(fn_name: str, kind: int, x: Tensor, other: Tensor, radius: Optional[Tensor]) -> Tensor:
    _no_grad_inputs(x, other)
    x, other = torch.broadcast_tensors(x, other)
    x, other = _f32c(x), _f32c(other)
    A = x.shape[-1]
    out = torch.empty_like(x)
    r = _radius_arg(kind, radius, x)
    check(getattr(load(), fn_name)(kind, ptr(x), ptr(other), ptr(out), x.numel() // A, _true_dim(kind, A), ptr(r),
                                   stream_ptr(x.device)))
    return out


def parallel_transport_mu0(kind: int, x: Tensor, dst: Tensor, radius: Optional[Tensor] = None) -> Tensor:
    return _pt("mvae_parallel_transport_mu0", kind, x, dst, radius)


def inverse_parallel_transport_mu0(kind: int, x: Tensor, src: Tensor, radius: Optional[Tensor] = None) -> Tensor:
    return _pt("mvae_inverse_parallel_transport_mu0", kind, x, src, radius)


def _at_rows(x: Tensor, at: Tensor, last: int) -> Tuple[Tensor, int]:
    """`at_point` may lack leading sample dims ([B,A] against [n,B,d]); the kernels index it modulo at_rows."""
    at = _f32c(at)
    lead = x.shape[:-1]
    if at.shape[:-1] == lead:
        return at, max(1, at.numel() // last)
    if at.dim() <= x.dim() and tuple(lead[len(lead) - (at.dim() - 1):]) == tuple(at.shape[:-1]):
        return at, max(1, at.numel() // last)
    at = at.expand(lead + (last,)).contiguous()
    return at, max(1, at.numel() // last)


def sample_projection_mu0(kind: int, v: Tensor, at_point: Tensor, radius: Optional[Tensor] = None):
    _no_grad_inputs(v, at_point)
    v = _f32c(v)
    d = v.shape[-1]
    A = ambient_dim(kind, d)
    at, at_rows = _at_rows(v, at_point, A)
    z = v.new_empty(v.shape[:-1] + (A,))
    u = torch.empty_like(z)
    r = _radius_arg(kind, radius, v)
    check(load().mvae_sample_projection_mu0(kind, ptr(v), ptr(at), ptr(z), ptr(u), v.numel() // d, at_rows, d, ptr(r),
                                            stream_ptr(v.device)))
    return z, (u, v)


def inverse_sample_projection_mu0(kind: int, z: Tensor, at_point: Tensor, radius: Optional[Tensor] = None):
    _no_grad_inputs(z, at_point)
    z = _f32c(z)
    A = z.shape[-1]
    d = _true_dim(kind, A)
    at, at_rows = _at_rows(z, at_point, A)
    u = torch.empty_like(z)
    v = z.new_empty(z.shape[:-1] + (d,))
    r = _radius_arg(kind, radius, z)
    check(load().mvae_inverse_sample_projection_mu0(kind, ptr(z), ptr(at), ptr(u), ptr(v), z.numel() // A, at_rows, d,
                                                    ptr(r), stream_ptr(z.device)))
    return u, v


def logdet(kind: int, u: Optional[Tensor], mu: Optional[Tensor], z: Optional[Tensor],
           radius: Optional[Tensor] = None) -> Tensor:
    _no_grad_inputs(u, mu, z)
    ref = _f32c(u if kind in (_lib.HYPERBOLOID, _lib.SPHERE) else z)
    A = ref.shape[-1]
    rows = ref.numel() // A
    out = ref.new_empty(ref.shape[:-1])
    r = _radius_arg(kind, radius, ref)
    mu_c, at_rows, z_c = None, rows, None
    if kind in _PROJECTED:
        z_c = ref
        mu_c, at_rows = _at_rows(ref, mu, A)
    check(load().mvae_logdet(kind, ptr(ref) if kind not in _PROJECTED else None, ptr(mu_c), ptr(z_c), ptr(out), rows,
                             at_rows, _true_dim(kind, A), ptr(r), stream_ptr(ref.device)))
    return out


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
    check(load().mvae_component_forward(layout.descs, layout.n, ptr(heads), heads.shape[-1], ptr(eps), layout.eps_dim,
                                        ptr(radii), ptr(z), layout.z_dim, ptr(kl), ptr(lq), ptr(lp), ptr(mu), ptr(sd),
                                        rows, head_rows, stream_ptr(heads.device)))
    return {"z": z, "kl": kl, "log_q": lq, "log_p": lp, "mu": mu, "std": sd}


def component_backward(layout: ComponentLayout, heads: Tensor, eps: Tensor, radii: Optional[Tensor], dz: Tensor,
                       dkl: Optional[Tensor], dkl_scalar: float = 0.0, want_dradii: bool = True):
    heads, eps, dz = _f32c(heads), _f32c(eps), _f32c(dz)
    rows = heads.shape[0]
    dheads = torch.zeros_like(heads)
    dradii = heads.new_zeros(layout.n) if want_dradii else None
    dkl = None if dkl is None else _f32c(dkl)
    radii = None if radii is None else _f32c(radii)
    check(load().mvae_component_backward(layout.descs, layout.n, ptr(heads), heads.shape[-1], ptr(eps), layout.eps_dim,
                                         ptr(radii), ptr(dz), layout.z_dim, ptr(dkl), float(dkl_scalar), ptr(dheads),
                                         ptr(dradii), rows, stream_ptr(heads.device)))
    return dheads, dradii


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


# --------------------------------------------------------------------------------------------- log-likelihood pieces
def bce_rows(logits: Tensor, x: Tensor) -> Tensor:
    """sum_j BCE-with-logits(logits[..., j], x[..., j]) with x broadcast over leading sample dims of logits."""
    logits, x = _f32c(logits), _f32c(x)
    D = logits.shape[-1]
    rows, x_rows = logits.numel() // D, x.numel() // D
    out = logits.new_empty(logits.shape[:-1])
    check(load().mvae_bce_rows(ptr(logits), ptr(x), ptr(out), rows, x_rows, D, stream_ptr(logits.device)))
    return out


def loglik_reduce(bce: Tensor, log_p: Tensor, log_q: Tensor):
    bce, log_p, log_q = _f32c(bce), _f32c(log_p), _f32c(log_q)
    n, B = bce.shape
    log_px, mi = bce.new_empty(B), bce.new_empty(B)
    check(load().mvae_loglik_reduce(ptr(bce), ptr(log_p), ptr(log_q), ptr(log_px), ptr(mi), n, B,
                                    stream_ptr(bce.device)))
    return log_px, mi
