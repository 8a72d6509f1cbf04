"""Deterministic synthetic inputs and initial states (no dataset / checkpoint exists on either box).

The recipes follow SURVEY.md section 8(d) / BASELINE.md section 3:
  * MNIST-like input: x_ij ~ Bernoulli(p_j), p = Generator(1234).rand(D)**3  (values {0,1}; mirrors the
    reference's dynamic binarisation, mt/data/image_reconstruction.py:44-53)
  * CIFAR-like input: x ~ U[0,1]^D (mirrors ToTensor, image_reconstruction.py:123-127)
  * eps ~ N(0,1)^{steps x B x sum(d)} from Generator(1000 + rank)
  * initial state: every parameter drawn U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the nn.Linear / nn.Conv2d default
    bound) from a generator keyed by crc32(parameter name) -- independent of module construction order, so the
    golden generator can `load_state_dict` the very same state into the reference model.

Everything is generated on the CPU with torch generators (bit-reproducible across boxes with the same torch build)
and moved to the device by the caller.
"""
import zlib
from typing import Dict, Iterable, Tuple

import torch


def pixel_probs(in_dim: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(1234)
    return torch.rand(in_dim, generator=g, dtype=torch.float64)**3


def binary_batches(steps: int, batch: int, in_dim: int, seed: int = 4321, dtype=torch.float32) -> torch.Tensor:
    """[steps, batch, in_dim] tensor of {0,1} -- a stand-in for dynamically binarised MNIST."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(steps, batch, in_dim, generator=g, dtype=torch.float64)
    return (u < pixel_probs(in_dim)).to(dtype)


def uniform_batches(steps: int, batch: int, in_dim: int, seed: int = 4321, dtype=torch.float32) -> torch.Tensor:
    """[steps, batch, in_dim] tensor in [0,1) -- a stand-in for CIFAR pixels (BCE with soft targets)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(steps, batch, in_dim, generator=g, dtype=torch.float64).to(dtype)


def eps_batches(steps: int, batch: int, total_true_dim: int, rank: int = 0, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator().manual_seed(1000 + rank)
    return torch.randn(steps, batch, total_true_dim, generator=g, dtype=torch.float64).to(dtype)


def _fan_in(name: str, shape: Tuple[int, ...], fan_in_of_weight: Dict[str, int]) -> int:
    if len(shape) >= 2:
        fi = 1
        for s in shape[1:]:
            fi *= s
        return fi
    return fan_in_of_weight.get(name.rsplit(".", 1)[0], 1)


def synthetic_state(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]], radius: float = 2.0,
                    dtype=torch.float32, transposed_conv: Iterable[str] = ()) -> Dict[str, torch.Tensor]:
    """Name-keyed deterministic state. Radii (`*_nradius`, `*_pradius`) are set to `radius`."""
    named_shapes = list(named_shapes)
    tconv = set(transposed_conv)
    fan = {}
    for name, shape in named_shapes:
        if len(shape) >= 2:
            mod = name.rsplit(".", 1)[0]
            if mod in tconv:  # ConvTranspose2d weight is [in, out, kh, kw]; torch's fan_in uses dim 1
                fi = shape[1]
                for s in shape[2:]:
                    fi *= s
            else:
                fi = 1
                for s in shape[1:]:
                    fi *= s
            fan[mod] = fi
    state = {}
    for name, shape in named_shapes:
        if name.endswith("radius") or name.endswith("_curvature"):
            state[name] = torch.tensor(radius, dtype=dtype)
            continue
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        bound = 1.0 / (fan[name.rsplit(".", 1)[0]]**0.5)
        u = torch.rand(tuple(shape), generator=g, dtype=torch.float64)
        state[name] = ((2.0 * u - 1.0) * bound).to(dtype)
    return state
