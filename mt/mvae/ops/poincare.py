"""mt/mvae/ops/poincare.py: PoincareBall and the module-level functions (:92-170).  The geoopt 0.1.0 arithmetic the
reference delegates to (`pm.*`) is restated in the kernels; value parity for it is pinned through the hyperboloid model
(DESIGN.md section 2)."""
from typing import Any, Tuple

import torch
from torch import Tensor

from mvae_amd import _lib, functional as _Fn
from mvae_amd.ops import PoincareBall  # noqa: F401

_K = _lib.POINCARE


def _c(radius: Tensor) -> Tensor:  # :108-109
    return 1 / radius**2


def _radius_of_c(c) -> Tensor:
    c = c if torch.is_tensor(c) else torch.tensor(float(c))
    return c.rsqrt() if c.dtype.is_floating_point else c.float().rsqrt()


def poincare_distance(x: Tensor, y: Tensor, radius: Tensor, keepdim: bool = True, **kwargs: Any) -> Tensor:  # :92-93
    return _Fn.geodesic_distance(_K, x, y, radius, keepdim=keepdim)


def poincare_distance_c(x: Tensor, y: Tensor, c: Tensor, keepdim: bool = True, **kwargs: Any) -> Tensor:  # :96-105
    return _Fn.geodesic_distance(_K, x, y, _radius_of_c(c), keepdim=keepdim)


def mu_0(shape: Tuple[int, ...], **kwargs: Any) -> Tensor:  # :112-113
    return torch.zeros(shape, **kwargs)


def parallel_transport_mu0(x: Tensor, dst: Tensor, radius: Tensor) -> Tensor:  # :116-117
    return _Fn.parallel_transport_mu0(_K, x, dst, radius)


def inverse_parallel_transport_mu0(x: Tensor, src: Tensor, radius: Tensor) -> Tensor:  # :120-121
    return _Fn.inverse_parallel_transport_mu0(_K, x, src, radius)


def exp_map(x: Tensor, at_point: Tensor, radius: Tensor) -> Tensor:  # :124-125
    return _Fn.exp_map(_K, x, at_point, radius)


def exp_map_c(x: Tensor, at_point: Tensor, c: Tensor) -> Tensor:  # :128-129
    return _Fn.exp_map(_K, x, at_point, _radius_of_c(c))


def exp_map_mu0(x: Tensor, radius: Tensor) -> Tensor:  # :132-133
    return _Fn.exp_map_mu0(_K, x, radius)


def exp_map_mu0_c(x: Tensor, c: Tensor) -> Tensor:  # :136-137
    return _Fn.exp_map_mu0(_K, x, _radius_of_c(c))


def inverse_exp_map(x: Tensor, at_point: Tensor, radius: Tensor) -> Tensor:  # :140-141
    return _Fn.inverse_exp_map(_K, x, at_point, radius)


def inverse_exp_map_c(x: Tensor, at_point: Tensor, c: Tensor) -> Tensor:  # :144-145
    return _Fn.inverse_exp_map(_K, x, at_point, _radius_of_c(c))


def inverse_exp_map_mu0(x: Tensor, radius: Tensor) -> Tensor:  # :148-149
    return _Fn.inverse_exp_map_mu0(_K, x, radius)


def sample_projection_mu0(x: Tensor, at_point: Tensor, radius: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:  # :152-157
    return _Fn.sample_projection_mu0(_K, x, at_point, radius)


def inverse_sample_projection_mu0(x_proj: Tensor, at_point: Tensor, radius: Tensor) -> Tuple[Tensor, Tensor]:  # :160-164
    return _Fn.inverse_sample_projection_mu0(_K, x_proj, at_point, radius)


def poincare_to_lorentz(y: Tensor, radius: Tensor) -> Tensor:  # :167-170
    return _Fn.manifold_aux(_lib.OP_TO_AMBIENT, _K, y, None, radius)


def lambda_x(x: Tensor, radius: Tensor, keepdim: bool = True) -> Tensor:
    """Conformal factor 2 / (1 - |x|^2 / R^2) (geoopt `pm.lambda_x`, used at poincare.py:154,163)."""
    out = _Fn.manifold_aux(_lib.OP_LAMBDA, _K, x, None, radius)
    return out if keepdim else out.squeeze(-1)


def mobius_add(x: Tensor, y: Tensor, radius: Tensor) -> Tensor:
    """x (+)_c y with c = 1/R^2 (geoopt `pm.mobius_add`, used at poincare.py:100)."""
    return _Fn.manifold_aux(_lib.OP_MOBADD, _K, x, y, radius)
