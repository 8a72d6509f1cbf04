"""mt/mvae/ops/spherical_projected.py: StereographicallyProjectedSphere and the module-level functions (:90-196)."""
from typing import Any, Tuple

import torch
from torch import Tensor

from mvae_amd import _lib, functional as _Fn
from mvae_amd.ops import StereographicallyProjectedSphere  # noqa: F401

_K = _lib.PROJ_SPHERE
MIN_NORM = 1e-15


def _c(radius: Tensor) -> Tensor:  # :116-117
    return 1 / radius**2


def _radius_of_K(K) -> Tensor:
    K = K if torch.is_tensor(K) else torch.tensor(float(K))
    return K.rsqrt()


def spherical_projected_distance(x: Tensor, y: Tensor, K: Tensor, keepdim: bool = True, **kwargs: Any) -> Tensor:  # :90-97
    return _Fn.geodesic_distance(_K, x, y, _radius_of_K(K), keepdim=keepdim)


def spherical_projected_gyro_distance(x: Tensor, y: Tensor, K: Tensor, keepdim: bool = True, **kwargs: Any) -> Tensor:  # :100-104
    return _Fn.geodesic_distance(_K, x, y, _radius_of_K(K), gyro=True, keepdim=keepdim)


def mob_add(x: Tensor, y: Tensor, K: Tensor) -> Tensor:  # :107-113
    return _Fn.manifold_aux(_lib.OP_MOBADD, _K, x, y, _radius_of_K(K))


def mu_0(shape: Tuple[int, ...], **kwargs: Any) -> Tensor:  # :120-121
    return torch.zeros(shape, **kwargs)


def lambda_x_c(x: Tensor, c: Tensor, dim: int = -1, keepdim: bool = True) -> Tensor:  # :124-125
    out = _Fn.manifold_aux(_lib.OP_LAMBDA, _K, x, None, _radius_of_K(c))
    return out if keepdim else out.squeeze(-1)


def lambda_x(x: Tensor, radius: Tensor, dim: int = -1, keepdim: bool = True) -> Tensor:  # :128-129
    out = _Fn.manifold_aux(_lib.OP_LAMBDA, _K, x, None, radius)
    return out if keepdim else out.squeeze(-1)


def parallel_transport_mu0(x: Tensor, dst: Tensor, radius: Tensor) -> Tensor:  # :140-141
    return _Fn.parallel_transport_mu0(_K, x, dst, radius)


def inverse_parallel_transport_mu0(x: Tensor, src: Tensor, radius: Tensor) -> Tensor:  # :144-145
    return _Fn.inverse_parallel_transport_mu0(_K, x, src, radius)


def exp_map(x: Tensor, at_point: Tensor, radius: Tensor) -> Tensor:  # :148-154
    return _Fn.exp_map(_K, x, at_point, radius)


def exp_map_mu0(x: Tensor, radius: Tensor) -> Tensor:  # :157-161
    return _Fn.exp_map_mu0(_K, x, radius)


def inverse_exp_map(x: Tensor, at_point: Tensor, radius: Tensor) -> Tensor:  # :164-169
    return _Fn.inverse_exp_map(_K, x, at_point, radius)


def inverse_exp_map_mu0(x: Tensor, radius: Tensor) -> Tensor:  # :172-175
    return _Fn.inverse_exp_map_mu0(_K, x, radius)


def sample_projection_mu0(x: Tensor, at_point: Tensor, radius: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:  # :178-181
    return _Fn.sample_projection_mu0(_K, x, at_point, radius)


def inverse_sample_projection_mu0(x_proj: Tensor, at_point: Tensor, radius: Tensor) -> Tuple[Tensor, Tensor]:  # :184-188
    return _Fn.inverse_sample_projection_mu0(_K, x_proj, at_point, radius)


def projected_to_spherical(y: Tensor, radius: Tensor) -> Tensor:  # :191-196
    return _Fn.manifold_aux(_lib.OP_TO_AMBIENT, _K, y, None, radius)
