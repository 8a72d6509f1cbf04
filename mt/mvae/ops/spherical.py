"""mt/mvae/ops/spherical.py: Sphere and the module-level functions with `radius=` (:58-133)."""
from typing import Any, Tuple

import torch
from torch import Tensor

from mvae_amd import _lib, functional as _Fn
from mvae_amd.ops import Sphere  # noqa: F401

_K = _lib.SPHERE


def _logdet(u: Tensor, radius: Tensor) -> Tensor:  # :58-67
    return _Fn.logdet(_K, u, None, None, radius)


def mu_0(shape: torch.Size, radius: Tensor, **kwargs: Any) -> Tensor:  # :70-71
    e = torch.zeros(shape, **kwargs)
    e[..., 0] = 1
    return e * radius


def parallel_transport_mu0(v: Tensor, dst: Tensor, radius: Tensor) -> Tensor:  # :74-77
    return _Fn.parallel_transport_mu0(_K, v, dst, radius)


def inverse_parallel_transport_mu0(x: Tensor, src: Tensor, radius: Tensor) -> Tensor:  # :80-83
    return _Fn.inverse_parallel_transport_mu0(_K, x, src, radius)


def exp_map(x: Tensor, at_point: Tensor, radius: Tensor) -> Tensor:  # :86-91
    return _Fn.exp_map(_K, x, at_point, radius)


def exp_map_mu0(x: Tensor, radius: Tensor) -> Tensor:  # :94-101: x is [..., d+1] with a leading zero
    return _Fn.exp_map_mu0(_K, x[..., 1:], radius)


def inverse_exp_map(x: Tensor, at_point: Tensor, radius: Tensor) -> Tensor:  # :104-109
    return _Fn.inverse_exp_map(_K, x, at_point, radius)


def inverse_exp_map_mu0(x: Tensor, radius: Tensor) -> Tensor:  # :112-116
    return _Fn.inverse_exp_map_mu0(_K, x, radius)


def sample_projection_mu0(x: Tensor, at_point: Tensor, radius: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:  # :119-123
    return _Fn.sample_projection_mu0(_K, x, at_point, radius)


def inverse_sample_projection_mu0(x: Tensor, at_point: Tensor, radius: Tensor) -> Tuple[Tensor, Tensor]:  # :126-129
    return _Fn.inverse_sample_projection_mu0(_K, x, at_point, radius)


def spherical_to_projected(x: Tensor, radius: Tensor) -> Tensor:  # :132-133
    return _Fn.manifold_aux(_lib.OP_TO_BALL, _K, x, None, radius)


def spherical_distance(x: Tensor, y: Tensor, radius: Tensor, keepdim: bool = True) -> Tensor:
    """R * acos(clamp(<x,y>/R^2, -1, 1)): the helper of tests/mvae/ops/test_spherical.py:45-48 as an operator."""
    return _Fn.geodesic_distance(_K, x, y, radius, keepdim=keepdim)
