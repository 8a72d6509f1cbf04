"""mt/mvae/ops/euclidean.py: Euclidean and its module-level functions (:62-99)."""
from typing import Any, Tuple

import torch
from torch import Tensor

from mvae_amd import _lib, functional as _Fn
from mvae_amd.ops import Euclidean  # noqa: F401

_K = _lib.EUCLIDEAN


def mu_0(shape: Tuple[int, ...], **kwargs: Any) -> Tensor:  # :62-63
    return torch.zeros(shape, **kwargs)


def parallel_transport_mu0(x: Tensor, dst: Tensor) -> Tensor:  # :66-67
    return _Fn.parallel_transport_mu0(_K, x, dst)


def inverse_parallel_transport_mu0(x: Tensor, src: Tensor) -> Tensor:  # :70-71
    return _Fn.inverse_parallel_transport_mu0(_K, x, src)


def exp_map(x: Tensor, at_point: Tensor) -> Tensor:  # :74-75
    return _Fn.exp_map(_K, x, at_point)


def exp_map_mu0(x: Tensor) -> Tensor:  # :78-79  (x / 2, sic)
    return _Fn.exp_map_mu0(_K, x)


def inverse_exp_map(x: Tensor, at_point: Tensor) -> Tensor:  # :82-83
    return _Fn.inverse_exp_map(_K, x, at_point)


def inverse_exp_map_mu0(x: Tensor) -> Tensor:  # :86-87
    return _Fn.inverse_exp_map_mu0(_K, x)


def sample_projection_mu0(x: Tensor, at_point: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:  # :90-93
    return _Fn.sample_projection_mu0(_K, x, at_point)


def inverse_sample_projection_mu0(x: Tensor, at_point: Tensor) -> Tuple[Tensor, Tensor]:  # :96-99
    return _Fn.inverse_sample_projection_mu0(_K, x, at_point)


def euclidean_distance(x: Tensor, y: Tensor, keepdim: bool = True) -> Tensor:
    """2 * |x - y|: the helper of tests/mvae/ops/test_euclidean.py:41-42 as an operator."""
    return _Fn.geodesic_distance(_K, x, y, keepdim=keepdim)
