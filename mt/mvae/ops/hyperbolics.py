"""mt/mvae/ops/hyperbolics.py: Hyperboloid and the module-level functions with `radius=` (:58-152).
`radius` is the radius itself (the reference's `self.radius`); the kernels clamp it to [1e-8, 1e8] like manifold.py:73-75."""
from typing import Any, Tuple

import torch
from torch import Tensor

from mvae_amd import _lib, functional as _Fn
from mvae_amd.ops import Hyperboloid  # noqa: F401

_K = _lib.HYPERBOLOID


def _logdet(u: Tensor, radius: Tensor) -> Tensor:  # :58-65
    return _Fn.logdet(_K, u, None, None, radius)


def mu_0(shape: Tuple[int, ...], radius: Tensor, **kwargs: Any) -> Tensor:  # :68-69
    e = torch.zeros(shape, **kwargs)
    e[..., 0] = 1
    return e * radius


def lorentz_product(x: Tensor, y: Tensor, keepdim: bool = False, dim: int = -1) -> Tensor:  # :72-78
    assert dim in (-1, x.dim() - 1), "the contraction runs over the coordinate (last) dimension"
    out = _Fn.manifold_aux(_lib.OP_LPROD, _K, x, y, 1.0)
    return out if keepdim else out.squeeze(-1)


def lorentz_norm(x: Tensor, **kwargs: Any) -> Tensor:  # :81-84
    out = _Fn.manifold_aux(_lib.OP_LNORM, _K, x, None, 1.0)
    return out if kwargs.get("keepdim", False) else out.squeeze(-1)


def parallel_transport_mu0(x: Tensor, dst: Tensor, radius: Tensor) -> Tensor:  # :87-93
    return _Fn.parallel_transport_mu0(_K, x, dst, radius)


def inverse_parallel_transport_mu0(x: Tensor, src: Tensor, radius: Tensor) -> Tensor:  # :96-103
    return _Fn.inverse_parallel_transport_mu0(_K, x, src, radius)


def exp_map(x: Tensor, at_point: Tensor, radius: Tensor) -> Tensor:  # :106-111
    return _Fn.exp_map(_K, x, at_point, radius)


def exp_map_mu0(x: Tensor, radius: Tensor) -> Tensor:  # :114-121: x is [..., d+1] with a leading zero (expand_proj_dims)
    return _Fn.exp_map_mu0(_K, x[..., 1:], radius)


def inverse_exp_map(x: Tensor, at_point: Tensor, radius: Tensor) -> Tensor:  # :124-128
    return _Fn.inverse_exp_map(_K, x, at_point, radius)


def inverse_exp_map_mu0(x: Tensor, radius: Tensor) -> Tensor:  # :131-135
    return _Fn.inverse_exp_map_mu0(_K, x, radius)


def sample_projection_mu0(x: Tensor, at_point: Tensor, radius: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:  # :138-142
    return _Fn.sample_projection_mu0(_K, x, at_point, radius)


def inverse_sample_projection_mu0(x: Tensor, at_point: Tensor, radius: Tensor) -> Tuple[Tensor, Tensor]:  # :145-148
    return _Fn.inverse_sample_projection_mu0(_K, x, at_point, radius)


def lorentz_to_poincare(x: Tensor, radius: Tensor) -> Tensor:  # :151-152
    return _Fn.manifold_aux(_lib.OP_TO_BALL, _K, x, None, radius)


def lorentz_distance(x: Tensor, y: Tensor, radius: Tensor, keepdim: bool = False) -> Tensor:
    """R * acosh(-<x,y>_L / R^2): the helper of tests/mvae/ops/test_hyperbolics.py:46-47 as an operator."""
    return _Fn.geodesic_distance(_K, x, y, radius, keepdim=keepdim)
