"""mt/mvae/ops/common.py: the guarded scalar functions with their custom backward rules, evaluated by the device code
of the manifold kernels (mvae_scalar_fn) and differentiable through torch.autograd."""
from typing import Any, Tuple

import torch

from mvae_amd import functional as _Fn

eps = 1e-8  # common.py:21-25
max_norm = 85
ln_2: torch.Tensor = torch.tensor(0.6931471805599453)
ln_1p2: torch.Tensor = ln_2 + 0.5
ln_2pi: torch.Tensor = ln_2 + 1.1447298858494002


def clamp(x: torch.Tensor, min: float = float("-inf"), max: float = float("+inf")) -> torch.Tensor:  # LeakyClamp :28-43
    return _Fn.guarded("clamp", x, min, max)


def atanh(x: torch.Tensor) -> torch.Tensor:  # :46-73
    return _Fn.guarded("atanh", x)


def acosh(x: torch.Tensor) -> torch.Tensor:  # :76-104
    return _Fn.guarded("acosh", x)


def cosh(x: torch.Tensor) -> torch.Tensor:  # :107-109
    return _Fn.guarded("cosh", x)


def sinh(x: torch.Tensor) -> torch.Tensor:  # :112-114
    return _Fn.guarded("sinh", x)


def sqrt(x: torch.Tensor) -> torch.Tensor:  # :117-119
    return _Fn.guarded("sqrt", x)


def logsinh(x: torch.Tensor) -> torch.Tensor:  # :122-128
    return _Fn.guarded("logsinh", x)


def logcosh(x: torch.Tensor) -> torch.Tensor:  # :131-136
    return _Fn.guarded("logcosh", x)


def e_i(i: int, shape: Tuple[int, ...], **kwargs: Any) -> torch.Tensor:  # :150-153 (allocation + one write)
    e = torch.zeros(shape, **kwargs)
    e[..., i] = 1
    return e


def expand_proj_dims(x: torch.Tensor) -> torch.Tensor:  # :156-158 (a concatenation with a zero column)
    return torch.cat((torch.zeros(x.shape[:-1] + (1,), device=x.device, dtype=x.dtype), x), dim=-1)
